// Internal interface between the dispatcher (fql_int4.hip) and the one-wave-per-SIMD GEMM's translation unit
// (fql_gemm_w4.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct FqlW4Args {
    const int8_t *limbs; const float *delta; const int32_t *rowsum;
    const uint8_t *packed; const float *scales; const float *zps;
    void *out; int out_kind;
    const int32_t *tpe; const int32_t *offs;
    int E, T, K, Kp, MBT, N;
    int n_tiles, m_slots, n_alt;
    float *scratch; const float *bias;
    long long blocks;
    hipStream_t stream;
};
// 0 on success, -1 launch failure, -2 no such instantiation
int fql_w4_launch(int L, int nf, int depth, const FqlW4Args &a);
int fql_w4_bn(int L, int nf);
