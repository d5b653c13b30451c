// Internal interface between the dispatcher (fql_int4.hip) and the one-wave-per-SIMD GEMM's translation unit
// (fql_gemm_w4.hip).  Not part of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// The one-launch form (pre-pass as the kernel's first phase): what the pre-pass needs beyond the GEMM's own arguments.
struct FqlW4Fused {
    const void *x;                      // float32 rows [n_src or T][K]
    const int32_t *gather;              // optional: grouped row t is row gather[t] of x
    int n_src;
    const float *row_weight;            // optional per-row output weight
    unsigned long long *flags;          // [ceil(T / 4)] one word per group of 4 grouped rows: == token once its limbs are in memory
    unsigned long long token;           // unique per launch (process salt << 32 | launch counter)
    int spin_limit;                     // polls before a workgroup quantises the rows it waits for itself
};

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
    bool fused = false;                 // fz is valid: launch the FUSED instantiation (no pre-pass launch before it)
    FqlW4Fused fz = {nullptr, nullptr, 0, nullptr, nullptr, 0ull, 0};
};
// 0 on success, -1 launch failure, -2 no such instantiation
int fql_w4_launch(int L, int nf, int depth, const FqlW4Args &a);
int fql_w4_bn(int L, int nf);
