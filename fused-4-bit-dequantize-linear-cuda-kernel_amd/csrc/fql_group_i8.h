// Per-GROUP scales along K on the INTEGER matrix cores (not in the reference, whose quantisation is
// per row: python/quantize.py:73-80).  scales / zps are [E][N][K / group]:
//   out[t][n] = sum_g scale[n][g] * ( sum_{k in g} q[n][k] x[t][k]  -  zp[n][g] * sum_{k in g} x[t][k] ).
// With x[t][k] = delta[t] * sum_l 256^l a_l[t][k] (the limbs of the activation pre-pass, csrc/fql_act_quant.h) both inner
// sums are exact integers per group: the first from v_mfma_i32_32x32x32_i8 started at zero for every group, the second
// from v_dot4 over the very fragment registers the MFMA reads.  At the end of a group the L integer accumulators of an
// output are folded in float32 and added, times the group's scale, to a float32 accumulator:
//   tot = sum_l 256^l * (P_l - zp * S_l)          f += scale * tot              out = delta[t] * f  (+ bias)
// (the per-row kernels do the same fold once, over the whole of K).  12 INT8 MFMAs per 128-k group and 32 x 32 block
// against 64 float32 ones in fql_group.h; the fold is ~9 VALU operations per output and group.
//
// Workgroup = 4 waves = 32 rows x 128 columns, one 32 x 32 block per wave (they share the activation fragments: L1 hits).
// Operands: activations straight from the fragment-native limb workspace (one coalesced 16-byte load per lane, limb and
// 32-k step); weights straight from global memory, 8 bytes per lane and step -- the bytes [8 b, 8 b + 8) of 16-byte chunk
// 2 v + (lane >> 5) of the row's 128-byte stage segment for step 2 v + b, the pairing the pre-pass's layout assumes --
// unpacked in registers.  Steps 2 v and 2 v + 1 together cover k = 64 v .. 64 v + 63 of a stage, so groups are multiples
// of 64.  Scales / zero points are read from a [E][G][N] transpose (one small kernel per call) as 16-byte loads.
#pragma once
#include "fql_common.h"

// in [E][N][G] -> out [E][G][N]
__global__ __launch_bounds__(256) void transpose_ng_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                           float *__restrict__ at, float *__restrict__ bt, int N, int G,
                                                           size_t total)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;           // index into the OUTPUT: coalesced writes
    if (i >= total) return;
    const size_t per = (size_t)N * G;
    const size_t e = i / per, r = i - e * per;
    const int g = (int)(r / N), n = (int)(r - (size_t)g * N);
    const size_t src = e * per + (size_t)n * G + g;
    at[i] = a[src];
    bt[i] = b[src];
}

template <int L>
__global__ __launch_bounds__(256) void group_i8_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales_t, const float *__restrict__ zps_t, float *__restrict__ out,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs, int E, int T, int K, int MBT, int N, int group,
    const float *__restrict__ bias, int has_res)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int e = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int row_lo = 0, cnt = T, pblk0 = 0;                              // rows of this expert, its first padded 32-row block
    if (tpe != nullptr) {
        int cp = 0, ct = 0;
        int lo_e = 0, cnt_e = 0, pad_e = 0;
        for (int base = 0; base < E; base += 64) {
            const ExpertLane xl = expert_chunk(tpe, offs, E, T, FQL_MB, base, lane, cp, ct);
            if (e >= base && e < base + 64) {
                lo_e = __shfl(xl.lo, e - base, 64);
                cnt_e = __shfl(xl.cnt, e - base, 64);
                pad_e = __shfl(xl.pad_excl, e - base, 64);
            }
        }
        row_lo = __builtin_amdgcn_readfirstlane(lo_e);
        cnt = __builtin_amdgcn_readfirstlane(cnt_e);
        pblk0 = __builtin_amdgcn_readfirstlane(pad_e) >> 5;
    }
    const int rb = blockIdx.y;
    if (rb * FQL_MB >= cnt) return;                                   // (uniform per workgroup)
    const int mb = pblk0 + rb;
    const int l31 = lane & 31, g2 = lane >> 5;
    const int n_blk = (int)blockIdx.x * 128 + wave * 32;
    if (n_blk >= N) return;
    const int n = n_blk + l31, nc = n < N ? n : N - 1;
    const int K2 = K >> 1, KB = K / FQL_KB, G = K / group, spg = group >> 5;     // 32-k steps per group (even)
    const uint8_t *wrow = packed + ((size_t)e * N + nc) * K2 + 16 * g2;
    const int8_t *abase = limbs + (size_t)mb * 8192 + lane * 16;
    const size_t a_stage = (size_t)MBT * 8192, a_limb = (size_t)KB * a_stage;

    // Heavy-tailed rows (csrc/fql_act_quant.h pass 3): a row block that holds one is walked a second time over the
    // residual limb set (limbs + L planes, quantum delta[T + t], 0 for rows without one) and the two results are added.
    const int rl = rb * FQL_MB + l31;
    const int t = row_lo + (rl < cnt ? rl : 0);
    const float d_main = delta[t];
    const float d_res = has_res ? delta[(size_t)T + t] : 0.0f;
    const int nset = __builtin_amdgcn_readfirstlane((__ballot(rl < cnt && d_res != 0.0f) != 0ull) ? 2 : 1);
    v16f res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = 0.0f;
  for (int set = 0; set < nset; ++set) {
    const int8_t *abase_s = abase + (size_t)set * L * a_limb;
    v16i acc[L];
    v16f f;
    int xs[L];
#pragma unroll
    for (int r = 0; r < 16; ++r) f[r] = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        xs[l] = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[l][r] = 0;
    }
    // steps are taken in pairs 2 v, 2 v + 1 (one 16-byte weight load feeds both: bytes [0, 8) and [8, 16) of chunk
    // 2 v + (lane >> 5)); one pair of loads ahead of the arithmetic
    auto load_a = [&](int pair, v4i (&dst)[2][L]) {
        const int kb = pair >> 2, v = pair & 3;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int l = 0; l < L; ++l)
                dst[b][l] = *reinterpret_cast<const v4i *>(abase_s + l * a_limb + kb * a_stage + (2 * v + b) * 1024);
    };
    auto load_w = [&](int pair) -> uint4 {
        const int kb = pair >> 2, v = pair & 3;
        return *reinterpret_cast<const uint4 *>(wrow + kb * (FQL_KB / 2) + 32 * v);
    };
    const int pairs = KB * 4;
    v4i an[2][L];
    load_a(0, an);
    uint4 wn = load_w(0);
    const float *st_e = scales_t + (size_t)e * G * N, *zt_e = zps_t + (size_t)e * G * N;
    const int n_q = n_blk + 4 * g2;                                   // register 4 q + c of this lane: column n_q + 8 q + c
    for (int pair = 0; pair < pairs; ++pair) {
        v4i ac[2][L];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int l = 0; l < L; ++l) ac[b][l] = an[b][l];
        const uint4 wc = wn;
        const int nxt = pair + 1 < pairs ? pair + 1 : pair;          // (clamped: loads are unconditional)
        load_a(nxt, an);
        wn = load_w(nxt);
        const uint32_t wd[4] = {wc.x, wc.y, wc.z, wc.w};
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            uint32_t lo0, hi0, lo1, hi1;
            unpack8(wd[2 * b], lo0, hi0);
            unpack8(wd[2 * b + 1], lo1, hi1);
            const v4i wf = v4i{(int)lo0, (int)hi0, (int)lo1, (int)hi1};
#pragma unroll
            for (int l = 0; l < L; ++l) {
                acc[l] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf, ac[b][l], acc[l], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) xs[l] = __builtin_amdgcn_sdot4(ac[b][l][i], 0x01010101, xs[l], false);
            }
        }
        const int step = 2 * pair + 1;
        if ((step + 1) % spg == 0) {                                  // ---- end of a group: fold into the float accumulator
            const int gi = step / spg;
            float xsf[L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                xsf[l] = (float)(xs[l] + __shfl_xor(xs[l], 32, 64));  // both k halves of this lane's token
                xs[l] = 0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nq = n_q + 8 * q;
                const int nqc = nq + 3 < N ? nq : (N >= 4 ? N - 4 : 0);           // (clamped: columns past N are not stored)
                float s4[4], z4[4];
                if ((N & 3) == 0) {
                    const v4f sv = *reinterpret_cast<const v4f *>(st_e + (size_t)gi * N + nqc);
                    const v4f zv = *reinterpret_cast<const v4f *>(zt_e + (size_t)gi * N + nqc);
#pragma unroll
                    for (int c = 0; c < 4; ++c) { s4[c] = sv[c]; z4[c] = zv[c]; }
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int nn = nq + c < N ? nq + c : N - 1;
                        s4[c] = st_e[(size_t)gi * N + nn];
                        z4[c] = zt_e[(size_t)gi * N + nn];
                    }
                }
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tot = 0.0f;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l) {
                        tot = fmaf(tot, 256.0f, fmaf(-z4[c], xsf[l], (float)acc[l][4 * q + c]));
                        acc[l][4 * q + c] = 0;
                    }
                    f[4 * q + c] = fmaf(s4[c], tot, f[4 * q + c]);
                }
            }
        }
    }
    const float dset = set == 0 ? d_main : d_res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = (set == 0) ? f[r] * dset : res[r] + f[r] * dset;
  }
    // ---- lane owns row t, registers 4 q .. 4 q + 3 are 4 consecutive columns
    if (rl >= cnt) return;
    const int n_q = n_blk + 4 * g2;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nq = n_q + 8 * q;
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            o[c] = res[4 * q + c];
            if (bias != nullptr && nq + c < N) o[c] += bias[(size_t)e * N + nq + c];
        }
        store_out4(out, 0, (size_t)t * N, nq, N, vec, o);
    }
#endif
}
