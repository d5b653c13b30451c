// Microbenchmark for the k-step structure of csrc/fql_gemm_w4.h: what, beside the 18 matrix instructions of a step, costs
// issue cycles at ONE wave per SIMD?  Knobs (template bit mask):
//   1  six ds_read_b128 per step, each reloading a weight-fragment register right after the 3 instructions that read it
//   2  the same reads, each issued one instruction group later (no write-after-read on a just-issued instruction)
//   4  three buffer_load_dwordx4 per step into an 8-step activation ring
//   8  the 6th fragment's accumulators in VGPRs through the instruction's VGPR form (inline assembly)
//  16  an s_cbranch around the 6th fragment (as the kernel has)
// hipcc --offload-arch=gfx950 -O3 -o w4_loop_probe w4_loop_probe.hip && ./w4_loop_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma_v(v16i &acc, const v4i &w, const v4i &a)
{
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(unsigned long long *cyc, int *sink, const int *gbuf, int iters, int nfr)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 98304 / 4; i += 256) reinterpret_cast<int *>(lds)[i] = (i * 2654435761u) & 0x0F0F0F0F;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gbuf, 0, 1 << 20, 0x00020000);
    v4i wf[6], afr[8][3];
    const int l31 = lane & 31, g = lane >> 5;
    const int rF0 = l31 * 256 + 16 * (g ^ (l31 & 15));
    for (int j = 0; j < 6; ++j) wf[j] = *reinterpret_cast<const v4i *>(lds + rF0 + j * 8192);
    for (int s = 0; s < 8; ++s) for (int l = 0; l < 3; ++l) afr[s][l] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (s * 3 + l) * 1024, 0);
    v16i acc[3][5], accv[3];
    for (int l = 0; l < 3; ++l) { for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[l][j][r] = 0; for (int r = 0; r < 16; ++r) accv[l][r] = 0; }
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const char *sb = lds + (it & 1) * 49152;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const char *fp = sb + (rF0 ^ (((ks + 1) & 7) * 32));
            const char *fc = sb + (rF0 ^ (ks * 32));
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                if (j == 5 && (MODE & 8)) {
                    if (!(MODE & 16) || j < nfr) {
#pragma unroll
                        for (int l = 0; l < 3; ++l) mfma_v(accv[l], wf[j], afr[ks][l]);
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < 3; ++l)
                        acc[l][j % 5] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], afr[ks][l], acc[l][j % 5], 0, 0, 0);
                }
                if (MODE & 1) wf[j] = *reinterpret_cast<const v4i *>(fp + j * 8192);
                if (MODE & 2) {
                    if (j == 0) wf[5] = *reinterpret_cast<const v4i *>(fc + 5 * 8192);
                    else wf[j - 1] = *reinterpret_cast<const v4i *>(fp + (j - 1) * 8192);
                }
            }
            if (MODE & 4) {
#pragma unroll
                for (int l = 0; l < 3; ++l) afr[ks][l] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, ((it & 7) * 24 + ks * 3 + l) * 1024, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    int s = 0;
    for (int l = 0; l < 3; ++l) { for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) s += acc[l][j][r]; for (int r = 0; r < 16; ++r) s += accv[l][r]; }
    for (int j = 0; j < 6; ++j) s += wf[j][0];
    for (int k = 0; k < 8; ++k) for (int l = 0; l < 3; ++l) s += afr[k][l][0];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char *name, int *gbuf)
{
    const int blocks = 256, iters = 2000;
    unsigned long long *cyc; int *sink;
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * 4);
    hipMalloc(&sink, sizeof(int) * blocks * 256);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 98304, 0, cyc, sink, gbuf, 10, 6);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 98304, 0, cyc, sink, gbuf, iters, 6);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[4];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-78s %7.1f cycles / k-step (18 matrix instructions = 576)  wall %.3f ms  clk %.2f GHz\n", name, (double)h[0] / (iters * 8.0), ms,
           (double)h[0] / (ms * 1e-3) / 1e9);
    hipFree(cyc); hipFree(sink);
}

int main()
{
    int *gbuf; hipMalloc(&gbuf, 1 << 20); hipMemset(gbuf, 0x11, 1 << 20);
    run<0>("18 matrix instructions, all AGPR accumulators (5 fragments, one twice)", gbuf);
    run<8>("15 AGPR + 3 VGPR-form (asm) accumulators", gbuf);
    run<8 | 16>("... + the branch around the VGPR-form fragment", gbuf);
    run<8 | 1>("15 + 3, six ds_read_b128 reloading a fragment right after its readers", gbuf);
    run<8 | 2>("15 + 3, six ds_read_b128 issued one group later", gbuf);
    run<8 | 4>("15 + 3, three buffer_load_dwordx4 per step (8-step ring)", gbuf);
    run<8 | 1 | 4>("15 + 3, ds_read right after readers + buffer loads", gbuf);
    run<8 | 2 | 4>("15 + 3, ds_read one group later + buffer loads", gbuf);
    run<8 | 16 | 1 | 4>("as the kernel: branch + ds_read right after readers + buffer loads", gbuf);
    run<1 | 4>("all AGPR (5 fragments), ds_read right after readers + buffer loads", gbuf);
    return 0;
}
