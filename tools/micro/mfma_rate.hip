// Microbenchmark: issue rate of v_mfma_i32_32x32x32_i8 / 16x16x64_i8 with operands in registers.
// hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NACC, bool RANDOM>
__global__ __launch_bounds__(512) void k32(unsigned long long *cyc, int *sink, int iters, int seed)
{
    v4i a, b;
    for (int i = 0; i < 4; ++i) { a[i] = RANDOM ? (threadIdx.x * 2654435761u + i * 40503u + seed) : 0; b[i] = RANDOM ? ((threadIdx.x * 97u + i * 7u + seed) & 0x0F0F0F0F) : 0; }
    v16i acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
__global__ __launch_bounds__(512) void k16(unsigned long long *cyc, int *sink, int iters, int seed)
{
    v4i a, b;
    for (int i = 0; i < 4; ++i) { a[i] = threadIdx.x * 2654435761u + i * 40503u + seed; b[i] = (threadIdx.x * 97u + i * 7u + seed) & 0x0F0F0F0F; }
    v4i acc[NACC];
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) acc[n][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[n], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int n = 0; n < NACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <typename F>
void run(const char *name, F launch, int blocks, int threads, int iters, int nacc, double ops_per_mfma)
{
    unsigned long long *cyc; int *sink;
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 8);
    hipMalloc(&sink, sizeof(int) * blocks * threads);
    launch(cyc, sink, 10);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    launch(cyc, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double nm = (double)iters * nacc;
    int waves = threads / 64;
    double total_ops = nm * ops_per_mfma * waves * blocks;
    printf("%-44s blocks=%4d waves/blk=%d: %.1f cyc/MFMA/wave (wave0)  wall %.3f ms  %.0f TOPS  implied clk %.2f GHz\n", name, blocks, waves,
           (double)h[0] / nm, ms, total_ops / (ms * 1e-3) / 1e12, (double)h[0] / (ms * 1e-3) / 1e9);
    hipFree(cyc); hipFree(sink);
}

int main()
{
    const int it = 20000;
    run("32x32x32 i8 zero data, 12 acc, 1 wave/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k32<12, false>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 65536.0);
    run("32x32x32 i8 random data, 12 acc, 1 wave/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k32<12, true>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 12, 65536.0);
    run("32x32x32 i8 random data, 9 acc, 2 waves/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k32<9, true>), dim3(256), dim3(512), 0, 0, c, s, n, 1); }, 256, 512, it, 9, 65536.0);
    run("32x32x32 i8 random data, 1 acc, 1 wave/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k32<1, true>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 1, 65536.0);
    run("32x32x32 i8 random, 12 acc, 1 wave/SIMD, 1 CU", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k32<12, true>), dim3(1), dim3(256), 0, 0, c, s, n, 1); }, 1, 256, it, 12, 65536.0);
    run("16x16x64 i8 random data, 16 acc, 1 wave/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k16<16>), dim3(256), dim3(256), 0, 0, c, s, n, 1); }, 256, 256, it, 16, 32768.0);
    run("16x16x64 i8 random data, 16 acc, 2 waves/SIMD", [&](auto c, auto s, int n) { hipLaunchKernelGGL((k16<16>), dim3(256), dim3(512), 0, 0, c, s, n, 1); }, 256, 512, it, 16, 32768.0);
    return 0;
}
