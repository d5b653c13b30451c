// Does the ORDER of the matrix instructions of a k-step change what the chip can sustain under its power limit?
// A step multiplies NF weight fragments (4-bit values in int8) with L activation limbs (full-range int8):
//   order J: for j: for l: acc[l][j] += W[j] x A[l]     -- the weight operand is held for L instructions, the activation
//                                                           operand (the high-entropy one) changes with every instruction
//   order L: for l: for j: acc[l][j] += W[j] x A[l]     -- the activation operand is held for NF instructions
// plus the same orders with non-negative activation bytes (0..127) and with zeros.  Operands live in registers, four
// k-steps' worth rotated; one wave per SIMD on every CU.  Wall time per launch = throughput under the power limit.
// hipcc --offload-arch=gfx950 -O3 -o mfma_order_probe mfma_order_probe.hip && ./mfma_order_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int ORDER, int DATA>      // DATA: 0 zeros, 1 random full-range activations, 2 activations 0..127, 3 activations: limb 2 small (|a| < 32) non-negative
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *cyc, int *sink, int iters, uint32_t seed)
{
    constexpr int NF = 5, L = 3, S = 4;
    v4i W[S][NF], A[S][L];
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    for (int s = 0; s < S; ++s) {
        for (int j = 0; j < NF; ++j)
            for (int i = 0; i < 4; ++i) W[s][j][i] = DATA == 0 ? 0 : (int)(hash32(tid * 977u + s * 131u + j * 17u + i + seed) & 0x0F0F0F0Fu);
        for (int l = 0; l < L; ++l)
            for (int i = 0; i < 4; ++i) {
                uint32_t r = hash32(tid * 7919u + s * 257u + l * 31u + i + seed * 3u);
                if (DATA == 0) r = 0;
                if (DATA == 2) r &= 0x7F7F7F7Fu;
                if (DATA == 3 && l == 2) r &= 0x1F1F1F1Fu;
                A[s][l][i] = (int)r;
            }
    }
    v16i acc[L][NF];
    for (int l = 0; l < L; ++l) for (int j = 0; j < NF; ++j) for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            if (ORDER == 0) {
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int l = 0; l < L; ++l) acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[s][j], A[s][l], acc[l][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j) acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(W[s][j], A[s][l], acc[l][j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int sum = 0;
    for (int l = 0; l < L; ++l) for (int j = 0; j < NF; ++j) for (int r = 0; r < 16; ++r) sum += acc[l][j][r];
    sink[tid] = sum;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int ORDER, int DATA>
double run(const char *name)
{
    const int blocks = 256, iters = 40000;      // 40000 x 60 matrix instructions ~ 40 ms
    unsigned long long *cyc; int *sink;
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    (void)hipMalloc(&sink, sizeof(int) * blocks * 256);
    hipLaunchKernelGGL((probe<ORDER, DATA>), dim3(blocks), dim3(256), 0, 0, cyc, sink, 2000, 1u);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f, sumt = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((probe<ORDER, DATA>), dim3(blocks), dim3(256), 0, 0, cyc, sink, iters, 7u + rep);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best; sumt += ms;
    }
    unsigned long long h[4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const double nm = (double)iters * 60.0;
    const double tops = nm * 65536.0 * 4 * blocks / (sumt / 5 * 1e-3) / 1e12;
    printf("%-64s %5.1f cyc/instr  wall %.2f ms (best %.2f)  %5.0f TOPS  clk %.2f GHz\n", name, (double)h[0] / nm, sumt / 5, best, tops,
           (double)h[0] / (sumt / 5 * 1e-3) / 1e9);
    (void)hipFree(cyc); (void)hipFree(sink);
    return sumt / 5;
}

int main()
{
    run<0, 0>("zeros, order J");
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 1>("random activations, order J (activation operand changes each)");
        run<1, 1>("random activations, order L (activation operand held for 5)");
        run<0, 2>("activations 0..127, order J");
        run<1, 2>("activations 0..127, order L");
        run<0, 3>("top limb 0..31, order J");
        run<1, 3>("top limb 0..31, order L");
    }
    return 0;
}
