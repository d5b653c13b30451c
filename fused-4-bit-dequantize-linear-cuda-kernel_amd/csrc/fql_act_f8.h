// Pre-pass of the fp8-activation path for rows that are ALREADY OCP e4m3 bytes (the caller quantised them, or an
// upstream kernel produced them): re-layout into the MFMA-fragment order of the limb workspace (one byte plane, the
// layout of csrc/fql_act_quant.h), per-row scale passed through, exact row sums for the zero-point fold.
//
//   out[t][n] = act_scale[t] * scale[n] * ( sum_k q[n][k] a[t][k]  -  zp[n] * sum_k a[t][k] ),   a = e4m3 value
//
// HBM-bound: reads T*K bytes, writes T*Kp bytes as full lines.  BASELINE.json configs[4] ("fp8 activations + INT4
// weights"); the reference lists FP8 as future work only (README.md:228).
#pragma once
#include "fql_act_quant.h"

__global__ __launch_bounds__(256) void act_f8_relayout_kernel(
    const uint8_t *__restrict__ xin, const float *__restrict__ act_scale, const int32_t *__restrict__ gather, int n_src,
    float *__restrict__ delta, int32_t *__restrict__ rowsum, int8_t *__restrict__ limbs, int T, int K, int KB, int MBT,
    int rblocks, void *__restrict__ out, int out_es, int N, const int32_t *__restrict__ tpe,
    const int32_t *__restrict__ offs, int E, int vec)
{
    __shared__ int s_tok[ACT_ROWS];
    __shared__ long long s_sum8[4][ACT_ROWS];
    __shared__ int s_nan[4][ACT_ROWS];
    __shared__ ActLookupShared s_lookup;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if ((int)blockIdx.x >= rblocks) {
        act_zero_uncovered((int)blockIdx.x - rblocks, out, out_es, N, tpe, offs, E, T);
        return;
    }
    const int p0 = blockIdx.x * ACT_ROWS;
    const int total = act_token_rows(p0, ACT_ROWS, s_tok, s_lookup, tpe, offs, E, T);
    if (p0 >= total) return;
    __syncthreads();
    const int r = tid & (ACT_ROWS - 1), col = tid / ACT_ROWS;
    const int tok = s_tok[r];
    const int p = p0 + r, mb = p >> 5, r32 = p & 31;
    const int src = tok >= 0 ? source_row(gather, n_src, tok) : 0;
    const uint8_t *xr = xin + (size_t)src * K;
    const int nch = KB * 16;                      // 16-byte chunks per padded row
    long long sum8 = 0;
    int nan = 0;
    for (int ch = col; ch < nch; ch += ACT_COLS) {
        const int k0 = ch * 16;
        uint32_t nat[4] = {0u, 0u, 0u, 0u};
        if (tok >= 0 && k0 < K) {
            if (vec) {                            // K % 16 == 0 and a 16-byte aligned base (host-checked)
                const v4i raw = *reinterpret_cast<const v4i *>(xr + k0);
                nat[0] = (uint32_t)raw[0]; nat[1] = (uint32_t)raw[1]; nat[2] = (uint32_t)raw[2]; nat[3] = (uint32_t)raw[3];
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    if (k0 + i < K) nat[i >> 2] |= (uint32_t)xr[k0 + i] << (8 * (i & 3));
            }
        }
        int part = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int v = (int)nat[q];
            // e4m3 NaN (S.1111.111) poisons the row, like a non-finite float32 activation does on the limb path
            nan |= (((nat[q] & 0x7Fu) == 0x7Fu) | ((nat[q] & 0x7F00u) == 0x7F00u) | ((nat[q] & 0x7F0000u) == 0x7F0000u) |
                    ((nat[q] & 0x7F000000u) == 0x7F000000u)) ? 1 : 0;
            part += (int)(__builtin_amdgcn_cvt_f32_fp8(v, 0) * 512.0f) + (int)(__builtin_amdgcn_cvt_f32_fp8(v, 1) * 512.0f)
                  + (int)(__builtin_amdgcn_cvt_f32_fp8(v, 2) * 512.0f) + (int)(__builtin_amdgcn_cvt_f32_fp8(v, 3) * 512.0f);
        }
        sum8 += part;
        const uint32_t w0 = __builtin_amdgcn_perm(nat[1], nat[0], 0x06040200u), w1 = __builtin_amdgcn_perm(nat[1], nat[0], 0x07050301u);
        const uint32_t w2 = __builtin_amdgcn_perm(nat[3], nat[2], 0x06040200u), w3 = __builtin_amdgcn_perm(nat[3], nat[2], 0x07050301u);
        const int kb = ch >> 4, c16 = ch & 15;
        const int ks = 2 * (c16 >> 2) + (c16 & 1), g = (c16 >> 1) & 1;
        int8_t *dst = limbs + (((size_t)kb) * MBT + mb) * 8192 + ((ks * 64) + g * 32 + r32) * 16;
        *reinterpret_cast<v4i *>(dst) = v4i{(int)w0, (int)w1, (int)w2, (int)w3};
    }
#pragma unroll
    for (int o = ACT_ROWS; o < 64; o <<= 1) { sum8 += __shfl_xor(sum8, o, 64); nan |= __shfl_xor(nan, o, 64); }
    if (lane < ACT_ROWS) { s_sum8[wave][lane] = sum8; s_nan[wave][lane] = nan; }
    __syncthreads();
    if (tid < ACT_ROWS && tok >= 0) {
        const long long tot = (s_sum8[0][tid] + s_sum8[1][tid]) + (s_sum8[2][tid] + s_sum8[3][tid]);
        const bool bad = (s_nan[0][tid] | s_nan[1][tid] | s_nan[2][tid] | s_nan[3][tid]) != 0;
        rowsum[tok] = __float_as_int((float)tot * 0x1p-9f);
        delta[tok] = bad ? __builtin_nanf("") : (act_scale != nullptr ? act_scale[src] : 1.0f);
    }
}
