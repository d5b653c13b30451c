// What does each kind of non-matrix instruction cost a single wave per SIMD between its matrix instructions?
// One k-step = 18 v_mfma_i32_32x32x32_i8 (six groups of three); per group the probe adds, in this order,
//   NDR ds_read_b128, NVM buffer_load_dwordx4, NVA v_and_b32 (dependent on nothing), NDW ds_write_b128, NSA s_add
// and reports shader cycles per step (576 = back-to-back matrix instructions).
// hipcc --offload-arch=gfx950 -O3 -o w4_issue_probe w4_issue_probe.hip && ./w4_issue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int NDR, int NVM, int NVA, int NDW, int NSA>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *cyc, int *sink, const int *gbuf, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) reinterpret_cast<int *>(lds)[i] = (i * 2654435761u) & 0x0F0F0F0F;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gbuf, 0, 1 << 20, 0x00020000);
    v4i wf[6], af[3], ld[6], vm[6];
    for (int j = 0; j < 6; ++j) { wf[j] = *reinterpret_cast<const v4i *>(lds + lane * 16 + j * 1024); ld[j] = wf[j]; vm[j] = wf[j]; }
    for (int l = 0; l < 3; ++l) af[l] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, l * 1024, 0);
    v16i acc[3][6];
    for (int l = 0; l < 3; ++l) for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;
    int va[8] = {lane, lane + 1, lane + 2, lane + 3, lane + 4, lane + 5, lane + 6, lane + 7};
    int sa = iters;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
#pragma unroll
            for (int l = 0; l < 3; ++l) acc[l][j % 5] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], af[l], acc[l][j % 5], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < NDR; ++n) ld[(j + n) % 6] = *reinterpret_cast<const v4i *>(lds + lane * 16 + ((j + n + it) & 31) * 1024);
#pragma unroll
            for (int n = 0; n < NVM; ++n) vm[(j + n) % 6] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((it + j + n) & 255) * 1024, 0);
#pragma unroll
            for (int n = 0; n < NVA; ++n) { va[n % 8] &= 0x0F0F0F0F + it; asm volatile("" : "+v"(va[n % 8])); }
#pragma unroll
            for (int n = 0; n < NDW; ++n) *reinterpret_cast<v4i *>(lds + 32768 + lane * 16 + ((j + n) & 15) * 1024) = ld[(j + n + 3) % 6];
#pragma unroll
            for (int n = 0; n < NSA; ++n) { sa += it; asm volatile("" : "+s"(sa)); }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = sa;
    for (int l = 0; l < 3; ++l) for (int j = 0; j < 6; ++j) for (int r = 0; r < 16; ++r) s += acc[l][j][r];
    for (int j = 0; j < 6; ++j) s += ld[j][0] + vm[j][0] + wf[j][1];
    for (int n = 0; n < 8; ++n) s += va[n];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int NDR, int NVM, int NVA, int NDW, int NSA>
void run(int *gbuf)
{
    const int blocks = 256, iters = 4000;
    unsigned long long *cyc; int *sink;
    (void)hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    (void)hipMalloc(&sink, sizeof(int) * blocks * 256);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe<NDR, NVM, NVA, NDW, NSA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipLaunchKernelGGL((probe<NDR, NVM, NVA, NDW, NSA>), dim3(blocks), dim3(256), 65536, 0, cyc, sink, gbuf, 10);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((probe<NDR, NVM, NVA, NDW, NSA>), dim3(blocks), dim3(256), 65536, 0, cyc, sink, gbuf, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[4];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("per group of 3 matrix instructions: %d ds_read_b128, %d buffer_load_x4, %2d VALU, %d ds_write_b128, %d SALU  -> %7.1f cycles / step (18 matrix instructions = 576)\n",
           NDR, NVM, NVA, NDW, NSA, (double)h[0] / iters);
    (void)hipFree(cyc); (void)hipFree(sink);
}

int main()
{
    int *gbuf; (void)hipMalloc(&gbuf, 1 << 20); (void)hipMemset(gbuf, 0x11, 1 << 20);
    run<0, 0, 0, 0, 0>(gbuf);
    run<1, 0, 0, 0, 0>(gbuf); run<2, 0, 0, 0, 0>(gbuf); run<4, 0, 0, 0, 0>(gbuf);
    run<0, 1, 0, 0, 0>(gbuf); run<0, 2, 0, 0, 0>(gbuf);
    run<0, 0, 3, 0, 0>(gbuf); run<0, 0, 6, 0, 0>(gbuf); run<0, 0, 12, 0, 0>(gbuf);
    run<0, 0, 0, 1, 0>(gbuf); run<0, 0, 0, 2, 0>(gbuf);
    run<0, 0, 0, 0, 3>(gbuf); run<0, 0, 0, 0, 6>(gbuf);
    run<1, 1, 0, 0, 0>(gbuf); run<1, 1, 3, 0, 0>(gbuf); run<1, 1, 3, 0, 3>(gbuf); run<1, 1, 3, 1, 3>(gbuf); run<2, 1, 6, 1, 3>(gbuf);
    return 0;
}
