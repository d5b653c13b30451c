// Translation unit of the one-wave-per-SIMD GEMM (fql_gemm_w4.h).  Built with -mllvm -amdgpu-mfma-vgpr-form so that the
// 288 accumulator registers of a wave may sit in AGPRs and VGPRs alike (Makefile); linked into libfql_int4.so.
#include "fql_gemm_w4.h"
#include "fql_w4_launch.h"

namespace {
struct PerDeviceFlagW4 { bool set[64] = {}; };
template <int L, int NF, int D, bool FUSED = false>
int launch_w4(const FqlW4Args &a)
{
    using C = W4Cfg<L, NF, D>;
    auto kern = gemm_w4_kernel<L, NF, D, FUSED>;
    static PerDeviceFlagW4 attr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr.set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES) != hipSuccess) return -1;
        attr.set[dev] = true;
    }
    (void)hipGetLastError();
    // (a cooperative launch of the same kernel, hipLaunchCooperativeKernel, cost +18 us per step: profiles/r03_ab_cooperative_launch.txt)
    hipLaunchKernelGGL(kern, dim3((unsigned)a.blocks), dim3(C::THREADS), C::LDS_BYTES, a.stream, a.limbs, a.delta, a.rowsum,
                       a.packed, a.scales, a.zps, a.out, a.out_kind, a.tpe, a.offs, a.E, a.T, a.K, a.Kp, a.MBT, a.N,
                       a.n_tiles, a.m_slots, a.scratch, a.bias, a.n_alt, a.fz);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
}  // namespace

int fql_w4_launch(int L, int nf, int depth, const FqlW4Args &a)
{
    if (L == 3 && nf == 6 && depth == 8 && !a.fused) return launch_w4<3, 6, 8>(a);
    if (L == 3 && nf == 6 && depth == 4) return a.fused ? launch_w4<3, 6, 4, true>(a) : launch_w4<3, 6, 4>(a);
    if (a.fused) return -2;
    return -2;
}
int fql_w4_bn(int L, int nf) { (void)L; return 32 * nf; }

#if defined(FQL_TRACE)
// Diagnostic builds only (tools/trace_w4.py): the stamps of wave 0 of the first eight workgroups.
extern "C" __attribute__((visibility("default"))) int fql_debug_trace_w4(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fql_trace_w4), sizeof(unsigned long long) * 8 * 64) == hipSuccess ? 0 : -1;
}
#endif
