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
// HBM-bound: reads T*K*4 bytes (second pass hits L2), writes L*T*Kp bytes.  One 256-thread
// workgroup per row.  For the MoE entry point the same launch also zero-fills the rows of `out`
// that no expert covers (reference semantics: torch::zeros, csrc/moe_int4_kernel.cu:109).
#pragma once
#include "fql_common.h"
#include <math.h>

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

// byte offset of the 8-byte group holding k0..k0+7 (k0 % 8 == 0) of padded row p, limb l
__device__ __forceinline__ size_t limb_offset(int l, int k0, int p, int KB, int MBT)
{
    const int kb = k0 >> 8, kin = k0 & 255;
    const int c = kin >> 5, v = c >> 1, g = c & 1, b = (kin >> 4) & 1;
    const int ks = 2 * v + b;
    const size_t blk = ((size_t)l * KB + kb) * MBT + (p >> 5);
    return ((blk * 8 + ks) * 64 + (g * 32 + (p & 31))) * 16 + (kin & 8);
}

template <int L>
__global__ __launch_bounds__(256) void act_quant_kernel(
    const float *__restrict__ x, int8_t *__restrict__ limbs, float *__restrict__ delta,
    int32_t *__restrict__ rowsum, int T, int K, int Kp, int MBT,
    float *__restrict__ out, int N, const int32_t *__restrict__ tpe,
    const int32_t *__restrict__ offs, int E)
{
    __shared__ float s_red[4];
    __shared__ int s_bad[4];
    __shared__ int s_sum[4 * L];
    const int t = blockIdx.x;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    int p = t;                            // padded row of t in the limb workspace
    if (tpe != nullptr) {                 // MoE: find the covering expert and its padded base
        bool covered = false;
        int pbase = 0;
        for (int e = 0; e < E; ++e) {
            int lo, cnt;
            expert_range(tpe, offs, e, T, lo, cnt);
            if (!covered && t >= lo && t < lo + cnt) {
                covered = true;
                p = pbase + (t - lo);
            }
            pbase += (cnt + FQL_MB - 1) / FQL_MB * FQL_MB;
        }
        if (!covered) {                   // rows covered by no expert are zeroed, not computed
            if (out != nullptr) {
                float *orow = out + (size_t)t * N;
                for (int i = tid; i < N; i += 256) orow[i] = 0.0f;
            }
            return;
        }
    }

    const float *xr = x + (size_t)t * K;
    const bool vec_ok = ((K & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);

    // pass 1: row max magnitude (+ non-finite detection)
    float m = 0.0f;
    int bad = 0;
    if (vec_ok) {
        const v4f *xv = reinterpret_cast<const v4f *>(xr);
        for (int i = tid; i < (K >> 2); i += 256) {
            v4f v = xv[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = fabsf(v[j]);
                bad |= !(a <= 3.402823466e+38f);
                m = fmaxf(m, a);
            }
        }
    } else {
        for (int i = tid; i < K; i += 256) {
            float a = fabsf(xr[i]);
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
    if (bad) m = 0.0f;                    // non-finite row: limbs 0, delta NaN -> the row's outputs are NaN

    const int e = act_exponent<L>(m);
    const float inv = ldexpf(1.0f, -e);
    const int KB = Kp / FQL_KB;

    // pass 2: quantise 8 consecutive k per thread, write one permuted 8-byte group per limb
    int sums[L];
#pragma unroll
    for (int l = 0; l < L; ++l) sums[l] = 0;

    const int G = Kp >> 3;
    for (int gi = tid; gi < G; gi += 256) {
        const int k0 = gi << 3;
        float v[8];
        if (vec_ok && k0 + 8 <= K) {
            v4f a = *reinterpret_cast<const v4f *>(xr + k0);
            v4f b = *reinterpret_cast<const v4f *>(xr + k0 + 4);
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
            v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (k0 + j < K) ? xr[k0 + j] : 0.0f;
        }
        uint32_t w[L][2];
#pragma unroll
        for (int l = 0; l < L; ++l) w[l][0] = w[l][1] = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int X = bad ? 0 : (int)rintf(v[j] * inv);
            const int pos = (j >> 1) + ((j & 1) << 2);          // (0,2,4,6,1,3,5,7) -> 0..7
#pragma unroll
            for (int l = 0; l < L; ++l) {
                int d;
                if (l == L - 1) d = X;                           // top limb: remaining value, in range by construction
                else { d = ((X + 128) & 255) - 128; X = (X - d) >> 8; }
                sums[l] += d;
                w[l][pos >> 2] |= (uint32_t)(d & 255) << ((pos & 3) * 8);
            }
        }
#pragma unroll
        for (int l = 0; l < L; ++l)
            *reinterpret_cast<uint2 *>(limbs + limb_offset(l, k0, p, KB, MBT)) = make_uint2(w[l][0], w[l][1]);
    }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        int s = wave_sum_i(sums[l]);
        if (lane == 0) s_sum[wave * L + l] = s;
    }
    __syncthreads();
    if (tid == 0) {
        delta[t] = bad ? __builtin_nanf("") : ldexpf(1.0f, e);
#pragma unroll
        for (int l = 0; l < L; ++l)
            rowsum[(size_t)l * T + t] = s_sum[l] + s_sum[L + l] + s_sum[2 * L + l] + s_sum[3 * L + l];
    }
}
