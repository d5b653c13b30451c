// Activation pre-pass of the MFMA path: float32 rows -> signed 8-bit limbs of a per-row
// fixed-point value, written in the layout the GEMM's matrix-core A operand wants.
//
//   x[t][k] ~= delta[t] * X[t][k],   X = sum_l 256^l * a_l,   a_l in [-128, 127]  (balanced digits)
//   delta[t] = 2^e, the smallest power of two with rint(max_k |x[t][k]| / 2^e) <= LIM(L)
//   rowsum[l][t] = sum_k a_l[t][k]      (folds the zero-point: sum_k (q-zp) a = sum q a - zp * rowsum)
//
// Limb layout ("fragment native"):  limbs[l][kb][mb][ks][lane][16 B]
//   kb  = k / 256                      weight-stage block (FQL_KB)
//   mb  = p / 32, p = padded row       expert e's rows start at pbase_e = sum_{e'<e} roundup(cnt_e', 32)
//   ks  = 0..7                         the 32-deep MFMA k-step inside the block
//   lane = g*32 + (p & 31)             exactly the lane that supplies row p to v_mfma_i32_32x32x32_i8
//   so one wave's A fragment for (l, kb, mb, ks) is 1 KiB contiguous: a single coalesced
//   buffer_load_dwordx4, no LDS.  Which 16 k a lane holds follows the weight side: lane group g of
//   step ks = 2v+b pairs with bytes [8b, 8b+8) of 16-byte chunk c = 2v+g of the weight row's 128-byte
//   stage segment, i.e. k = 32c + 16b + [0,16) of the block; inside every aligned group of 8 the order
//   is (0,2,4,6,1,3,5,7), the order the in-register nibble unpack (unpack8) produces.
//   Columns K..Kp-1 are zero.
//
// HBM-bound: reads T*K*4 bytes twice (the second read hits L2 / Infinity Cache), writes L*T*Kp bytes
// in contiguous 8 KiB blocks.  Two launches: per-row scale, then the tiled conversion.
#pragma once
#include "fql_common.h"
#include <math.h>

// Fused dispatch: when `gather` is given, grouped row t is read from token row
// gather[t] of x (the sort-by-expert permutation of routing.py:117-149), so the [T*top_k, K] gathered
// copy of the activations is never materialised.  Indices are clamped into [0, n_src).
__device__ __forceinline__ int source_row(const int32_t *gather, int n_src, int t)
{
    if (gather == nullptr) return t;
    int s = gather[t];
    s = s < 0 ? 0 : s;
    return s < n_src ? s : n_src - 1;
}

template <int L> struct LimbLimit;
template <> struct LimbLimit<1> { static constexpr int value = 127; };
template <> struct LimbLimit<2> { static constexpr int value = 127 * 256 + 127; };
template <> struct LimbLimit<3> { static constexpr int value = 127 * 65536 + 127 * 256 + 127; };

template <int L>
__device__ __forceinline__ int act_exponent(float m)
{
    if (m == 0.0f) return 0;
    int ex;
    (void)frexpf(m, &ex);                 // m = f * 2^ex, f in [0.5, 1)
    int e = (ex - 1) - (8 * L - 2);       // m * 2^-e in [2^(8L-2), 2^(8L-1))
    e = e < -126 ? -126 : e;              // keep 2^e and 2^-e normal
    if (rintf(m * ldexpf(1.0f, -e)) > (float)LimbLimit<L>::value) e += 1;
    return e;
}

// ---- kernel 1: per-row scale.  One 256-thread workgroup per row (all loads independent and in
//      flight together): delta[t] = 2^e (NaN for a non-finite row) and the per-limb digit sums rowsum[l][t];
//      for the MoE entry point also zero-fills the rows of `out` that no expert covers (reference
//      semantics: torch::zeros, csrc/moe_int4_kernel.cu:109).
template <int L>
__global__ __launch_bounds__(256) void act_scale_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ gather, int n_src, float *__restrict__ delta,
    int32_t *__restrict__ rowsum, int T, int K,
    float *__restrict__ out, int N, const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs, int E)
{
    __shared__ float s_red[4];
    __shared__ int s_bad[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t = blockIdx.x;
    if (tpe != nullptr) {
        bool covered = false;
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) {
            const ExpertLane xl = expert_chunk(tpe, offs, E, T, FQL_MB, base, lane, cp, ct);
            covered |= __ballot(t >= xl.lo && t < xl.lo + xl.cnt) != 0ull;
        }
        if (!covered) {
            if (out != nullptr) {
                float *orow = out + (size_t)t * N;
                for (int i = tid; i < N; i += 256) orow[i] = 0.0f;
            }
            return;
        }
    }
    const float *xr = x + (size_t)source_row(gather, n_src, t) * K;
    const bool vec_ok = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    float m = 0.0f;
    int bad = 0;
    if (vec_ok) {
        const v4f *xv = reinterpret_cast<const v4f *>(xr);
        const int nv = K >> 2;
        for (int i0 = 0; i0 < nv; i0 += 1024) {               // 4 independent 16-byte loads per thread per trip
            v4f v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                v[u] = (i < nv) ? xv[i] : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = fabsf(v[u][j]);
                    bad |= !(a <= 3.402823466e+38f);
                    m = fmaxf(m, a);
                }
        }
    } else {
        for (int i = tid; i < K; i += 256) {
            const float a = fabsf(xr[i]);
            bad |= !(a <= 3.402823466e+38f);
            m = fmaxf(m, a);
        }
    }
    m = wave_max(m);
    bad = __any(bad) ? 1 : 0;
    if (lane == 0) { s_red[wave] = m; s_bad[wave] = bad; }
    __syncthreads();
    m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
    bad = s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3];
    const int e = act_exponent<L>(bad ? 0.0f : m);
    const float inv = bad ? 0.0f : ldexpf(1.0f, -e);

    // second pass over the row (L1 / L2 hits): the per-limb digit sums, so kernel 2 needs no atomics
    int sums[L];
#pragma unroll
    for (int l = 0; l < L; ++l) sums[l] = 0;
    auto add = [&](float v) {
        int X = (int)rintf(v * inv);
#pragma unroll
        for (int l = 0; l < L; ++l) {
            int d;
            if (l == L - 1) d = X;
            else { d = ((X + 128) & 255) - 128; X = (X - d) >> 8; }
            sums[l] += d;
        }
    };
    if (vec_ok) {
        const v4f *xv = reinterpret_cast<const v4f *>(xr);
        const int nv = K >> 2;
        for (int i0 = 0; i0 < nv; i0 += 1024) {
            v4f v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * 256 + tid;
                v[u] = (i < nv) ? xv[i] : v4f{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) add(v[u][j]);
        }
    } else {
        for (int i = tid; i < K; i += 256) add(xr[i]);
    }
    __shared__ int s_sum[4 * L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const int sv = wave_sum_i(sums[l]);
        if (lane == 0) s_sum[wave * L + l] = sv;
    }
    __syncthreads();
    if (tid == 0) {
        delta[t] = bad ? __builtin_nanf("") : ldexpf(1.0f, e);
#pragma unroll
        for (int l = 0; l < L; ++l)
            rowsum[(size_t)l * T + t] = s_sum[l] + s_sum[L + l] + s_sum[2 * L + l] + s_sum[3 * L + l];
    }
}

// ---- kernel 2: one workgroup per (32-row block mb, 256-k block kb).  Coalesced float4 reads of the
//      32 x 256 tile (all 8 loads of a thread issued before any is used), quantise to L limbs, scatter the
//      bytes into the fragment image in LDS, then stream the L contiguous 8 KiB fragment blocks out with
//      16-byte stores.
template <int L>
__global__ __launch_bounds__(256) void act_limbs_kernel(
    const float *__restrict__ x, const int32_t *__restrict__ gather, int n_src,
    const float *__restrict__ delta, int8_t *__restrict__ limbs, int T, int K, int KB, int MBT,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs, int E)
{
    __shared__ __attribute__((aligned(16))) char img[L * 8192];
    __shared__ int s_tok[FQL_MB];
    const int mb = blockIdx.x, kb = blockIdx.y;
    const int tid = threadIdx.x;
    const int gi = tid & 31;                      // 8-group inside the 256-k block
    const bool vec_ok = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

    // token row of each of the block's 32 padded rows (-1: padding)
    if (tpe == nullptr) {
        if (tid < FQL_MB) s_tok[tid] = (mb * FQL_MB + tid < T) ? mb * FQL_MB + tid : -1;
    } else {
        // wave 0 publishes the expert table chunk by chunk; 32 threads then look their row up in parallel
        __shared__ int s_lo[64], s_cnt[64], s_pad[64];
        int t_found = -1;
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) {
            if (tid < 64) {
                const ExpertLane xl = expert_chunk(tpe, offs, E, T, FQL_MB, base, tid, cp, ct);
                s_lo[tid] = xl.lo; s_cnt[tid] = xl.cnt; s_pad[tid] = xl.pad_excl;
            }
            __syncthreads();
            if (tid < FQL_MB && t_found < 0) {
                const int p = mb * FQL_MB + tid;
                const int ne = (E - base) < 64 ? (E - base) : 64;
                for (int i = 0; i < ne; ++i) {
                    const int rel = p - s_pad[i];
                    if (rel >= 0 && rel < s_cnt[i]) { t_found = s_lo[i] + rel; break; }
                }
            }
            __syncthreads();
        }
        if (tid < FQL_MB) s_tok[tid] = t_found;
    }
    __syncthreads();

    const int k0 = kb * FQL_KB + gi * 8;
    int tok[4];
    float inv[4];
    v4f va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        tok[j] = s_tok[(tid >> 5) + 8 * j];
        va[j] = vb[j] = v4f{0.f, 0.f, 0.f, 0.f};
        inv[j] = 0.0f;
        if (tok[j] >= 0) {
            const float *xr = x + (size_t)source_row(gather, n_src, tok[j]) * K;
            if (vec_ok && k0 + 8 <= K) {
                va[j] = *reinterpret_cast<const v4f *>(xr + k0);
                vb[j] = *reinterpret_cast<const v4f *>(xr + k0 + 4);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    va[j][i] = (k0 + i < K) ? xr[k0 + i] : 0.0f;
                    vb[j][i] = (k0 + 4 + i < K) ? xr[k0 + 4 + i] : 0.0f;
                }
            }
            const float d = delta[tok[j]];
            inv[j] = (d == d) ? 1.0f / d : 0.0f;  // delta is a power of two: exact reciprocal; NaN row -> limbs 0
        }
    }
    // fragment position of this 8-group: chunk c = gi / 4 -> (v = c >> 1, g = c & 1), half b, byte 8*(gi & 1)
    const int c = gi >> 2, b = (gi >> 1) & 1;
    const int ks = 2 * (c >> 1) + b, g = c & 1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (tid >> 5) + 8 * j;
        const float v[8] = {va[j][0], va[j][1], va[j][2], va[j][3], vb[j][0], vb[j][1], vb[j][2], vb[j][3]};
        uint32_t w[L][2];
#pragma unroll
        for (int l = 0; l < L; ++l) w[l][0] = w[l][1] = 0u;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int X = (int)rintf(v[i] * inv[j]);
            const int pos = (i >> 1) + ((i & 1) << 2);           // (0,2,4,6,1,3,5,7) -> 0..7
#pragma unroll
            for (int l = 0; l < L; ++l) {
                int d;
                if (l == L - 1) d = X;                            // top limb: remaining value, in range by construction
                else { d = ((X + 128) & 255) - 128; X = (X - d) >> 8; }
                w[l][pos >> 2] |= (uint32_t)(d & 255) << ((pos & 3) * 8);
            }
        }
        const int off = ((ks * 64) + g * 32 + r) * 16 + ((gi & 1) << 3);
#pragma unroll
        for (int l = 0; l < L; ++l) *reinterpret_cast<uint2 *>(img + l * 8192 + off) = make_uint2(w[l][0], w[l][1]);
    }
    __syncthreads();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        int8_t *dst = limbs + (((size_t)l * KB + kb) * MBT + mb) * 8192;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            *reinterpret_cast<v4i *>(dst + (i * 256 + tid) * 16) = *reinterpret_cast<const v4i *>(img + l * 8192 + (i * 256 + tid) * 16);
    }
}
