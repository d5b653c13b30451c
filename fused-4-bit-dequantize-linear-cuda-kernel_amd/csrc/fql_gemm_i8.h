// Grouped INT4-weight x INT8-limb-activation GEMM on the CDNA4 matrix cores
// (v_mfma_i32_32x32x32_i8), one launch for all experts.  The linear op is the 1-group case.
//
//   out[t][n] = scale[e][n] * delta[t] * sum_l 256^l * ( sum_k q[e][n][k] * a_l[t][k]  -  zp[e][n] * rowsum_l[t] )
//
// for every row t of expert e's range.  The inner integer dot products are exact (i32
// accumulation), so results do not depend on tile shape, K order or which GPU ran the row.
//
// Data movement per workgroup (BM = 32*MF*WM rows, BN = 32*NF*WN output columns, BK = 64):
//   * packed weights: HBM -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds), each byte once per
//     workgroup; read back with one ds_read_b128 per 32x64 fragment; nibbles are unpacked in
//     registers with 3 VALU ops per 8 weights (unpack8) straight into the MFMA B operand.
//   * activation limbs: L2 -> LDS by LDS-DMA; ds_read_b128 per 32x32 A fragment.
//   * LDS images are XOR-swizzled on the DMA *source* side (the LDS destination of an LDS-DMA is
//     lane-linear) so that both fragment reads are bank-conflict-free.
//   * double-buffered stages, one workgroup barrier per stage; loads of stage s+1 fly under the
//     MFMAs of stage s.
//
// Replaces (reference, CUDA): csrc/quantized_linear_kernel.cu:90-279 (one thread per output,
// weights re-read per batch row) and csrc/moe_int4_kernel.cu:17-136 (one <<<1,256>>> launch and two
// host syncs per expert).
#pragma once
#include "fql_common.h"

template <int L, int WM, int WN, int MF, int NF>
struct GemmCfg {
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = 32 * MF * WM;
    static constexpr int BN = 32 * NF * WN;
    static constexpr int A_BYTES = L * BM * FQL_BK;          // per stage
    static constexpr int B_BYTES = BN * (FQL_BK / 2);        // per stage
    static constexpr int STAGE = A_BYTES + B_BYTES;
    static constexpr int CA = A_BYTES / 1024;                // 1 KiB LDS-DMA pieces: 16 rows x 64 B
    static constexpr int CB = B_BYTES / 1024;                // 32 rows x 32 B
    static constexpr int CPW = (CA + CB + NW - 1) / NW;      // pieces per wave per stage
    static_assert(BM % 16 == 0 && BN % 32 == 0, "tile must be whole LDS-DMA pieces");
};

template <int L, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(64 * WM * WN) void gemm_i8_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, float *__restrict__ out,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int N, int n_tiles, int m_slots)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub (hipcc drops the
                                      // stub of a template kernel whose body holds LDS-DMA builtins)
    using C = GemmCfg<L, WM, WN, MF, NF>;
    __shared__ __attribute__((aligned(16))) char lds[2 * C::STAGE];

    // ---- which tile: (m-slot, n-tile), m-slot major so that workgroups on one XCD share an
    //      expert's activation panel in that XCD's L2 while each weight byte streams once.
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int ms = tile / n_tiles;
    const int nt = tile - ms * n_tiles;

    int e = 0, row0 = 0, rows_valid = 0;
    if (tpe == nullptr) {                                   // linear: one group covering all T rows
        row0 = ms * C::BM;
        rows_valid = T - row0;
    } else {                                                // MoE: offsets/counts read on the device
        int run = 0;
        bool found = false;
        for (int i = 0; i < E; ++i) {
            long long lo = offs[i], hi = lo + (long long)tpe[i];
            lo = lo < 0 ? 0 : lo;
            hi = hi > T ? T : hi;
            const int cnt = hi > lo ? (int)(hi - lo) : 0;
            const int tiles = (cnt + C::BM - 1) / C::BM;
            if (!found && ms < run + tiles) {
                found = true;
                e = i;
                row0 = (int)lo + (ms - run) * C::BM;
                rows_valid = cnt - (ms - run) * C::BM;
            }
            run += tiles;
        }
        if (!found) return;
    }
    if (rows_valid <= 0) return;
    if (rows_valid > C::BM) rows_valid = C::BM;
    const int n0 = nt * C::BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, g = lane >> 5;

    // ---- buffer descriptors (bounds-checked: rows past T / N and the K tail read as zero)
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)L * T * Kp), 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);

    // ---- per-lane source offsets of this wave's LDS-DMA pieces (constant over the K loop)
    int voff[C::CPW];
#pragma unroll
    for (int i = 0; i < C::CPW; ++i) {
        const int c = wave + i * C::NW;
        if (c < C::CA) {                                    // 16 rows x 64 B of one limb plane
            const int l = c / (C::BM / 16), cj = c - l * (C::BM / 16);
            const int rr = lane >> 2, cs = lane & 3;
            const int lc = cs ^ ((rr >> 2) & 3);            // source-side swizzle
            voff[i] = (l * T + row0 + cj * 16 + rr) * Kp + lc * 16;
        } else {                                            // 32 weight rows x 32 B
            const int cb = c - C::CA;
            const int nn = lane >> 1, hs = lane & 1;
            const int lh = hs ^ ((nn >> 3) & 1);
            voff[i] = (n0 + cb * 32 + nn) * (K >> 1) + lh * 16;
        }
    }

#define FQL_STAGE_LOAD(kt_, buf_)                                                                          \
    do {                                                                                                   \
        char *base_ = lds + (buf_) * C::STAGE;                                                             \
        _Pragma("unroll") for (int i = 0; i < C::CPW; ++i) {                                               \
            const int c = wave + i * C::NW;                                                                \
            if (c < C::CA)                                                                                 \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(base_ + c * 1024), 16, voff[i],      \
                                                         (kt_) * FQL_BK, 0, 0);                            \
            else if (c < C::CA + C::CB)                                                                    \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(base_ + c * 1024), 16, voff[i],      \
                                                         (kt_) * (FQL_BK / 2), 0, 0);                      \
        }                                                                                                  \
    } while (0)

    v16i acc[L][MF][NF];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[l][i][j][r] = 0;

    // ---- fragment read offsets inside a stage
    int a_off[MF], a_sw[MF], b_off[NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) {
        const int r = (wm * MF + i) * 32 + l31;
        a_off[i] = r * FQL_BK;
        a_sw[i] = (r >> 2) & 3;
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = (wn * NF + j) * 32 + l31;
        b_off[j] = C::A_BYTES + n * (FQL_BK / 2) + 16 * (g ^ ((n >> 3) & 1));
    }

    const int KT = (K + FQL_BK - 1) / FQL_BK;
    FQL_STAGE_LOAD(0, 0);
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                       // stage kt landed; everyone is done with the other buffer
        if (kt + 1 < KT) FQL_STAGE_LOAD(kt + 1, (kt + 1) & 1);
        const char *sb = lds + (kt & 1) * C::STAGE;

        v4i braw[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) braw[j] = *reinterpret_cast<const v4i *>(sb + b_off[j]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two 32-deep MFMA k-steps per 64-deep stage
            v4i bfr[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                uint32_t lo0, hi0, lo1, hi1;
                unpack8((uint32_t)braw[j][2 * s], lo0, hi0);
                unpack8((uint32_t)braw[j][2 * s + 1], lo1, hi1);
                bfr[j][0] = (int)lo0; bfr[j][1] = (int)hi0; bfr[j][2] = (int)lo1; bfr[j][3] = (int)hi1;
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int i = 0; i < MF; ++i) {
                    const v4i afr = *reinterpret_cast<const v4i *>(
                        sb + l * (C::BM * FQL_BK) + a_off[i] + 16 * ((2 * g + s) ^ a_sw[i]));
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[l][i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(afr, bfr[j], acc[l][i][j], 0, 0, 0);
                }
        }
    }

    // ---- epilogue: fold zero-point, combine limbs, scale.  C/D layout of the 32x32 MFMA:
    //      col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    float sc[NF], zp[NF];
    int col[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        col[j] = n0 + (wn * NF + j) * 32 + l31;
        const bool ok = col[j] < N;
        sc[j] = ok ? scales[(size_t)e * N + col[j]] : 0.0f;
        zp[j] = ok ? zps[(size_t)e * N + col[j]] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = (wm * MF + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * g;
            if (rl >= rows_valid) continue;
            const int t = row0 + rl;
            const float d = delta[t];
            float rs[L];
#pragma unroll
            for (int l = 0; l < L; ++l) rs[l] = (float)rowsum[(size_t)l * T + t];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                if (col[j] >= N) continue;
                float tot = 0.0f;
#pragma unroll
                for (int l = L - 1; l >= 0; --l) {
                    const float c = fmaf(-zp[j], rs[l], (float)acc[l][i][j][r]);
                    tot = fmaf(tot, 256.0f, c);
                }
                out[(size_t)t * N + col[j]] = (tot * d) * sc[j];
            }
        }
#undef FQL_STAGE_LOAD
#endif  // __HIP_DEVICE_COMPILE__
}
