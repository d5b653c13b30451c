// Grouped INT4 x INT8-limb GEMM for SHORT row groups (decode-size batches, few rows per expert, a single
// expert's 128 rows on one GPU of an expert-parallel job): the weight stream dominates, and a 128 x 192 tile
// per workgroup leaves most CUs idle or multiplies zeros.  Here a tile is ONE 32-row MFMA block; the eight
// waves of a workgroup split the tile's K range KG ways and its columns 8/KG ways, so that
//   * the tile count is high enough to give every CU work even for one expert (a CU = 32 rows x BN columns),
//   * each wave streams ITS OWN weight rows: 8 rows x 128 B = 8 full cache lines per instruction, global ->
//     VGPR -> a wave-private LDS slab (XOR-swizzled) -> MFMA fragments.  No weight byte is shared between
//     waves, so there is no workgroup barrier anywhere in the K loop -- every wave runs at its own pace with
//     a full stage (NF x 4 KiB) of HBM loads in flight,
//   * the activation fragments come straight from the fragment-native limb workspace (L2 hits), as in the
//     wide kernel,
//   * the KG partial accumulators are added through LDS at the end (int32: exact, order-free), then the
//     same epilogue as the wide kernel.  Results are bit-identical to every other tile configuration.
#pragma once
#include "fql_common.h"
#include "fql_gemm_i8.h"

template <int L, int NF, int KG, int DEPTH>
struct Rows32Cfg {
    static constexpr int NW = 8;
    static constexpr int NG = NW / KG;                        // column groups of the workgroup
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = FQL_MB;
    static constexpr int BN = 32 * NF * NG;
    static constexpr int KS = FQL_KB / 32;
    static constexpr int D = DEPTH;
    static constexpr int PIECES = NF * 4;                     // 1 KiB weight pieces per wave per 256-k stage
    static constexpr int SLAB = NF * 32 * (FQL_KB / 2);       // wave-private LDS bytes (one stage of packed weights)
    static constexpr int ACC_BYTES = L * NF * 16 * 64 * 4;    // one wave's accumulators
    static constexpr int RED_BYTES = (KG > 1) ? (KG / 2) * NG * ACC_BYTES : 0;   // first round of the K-group tree
    static constexpr int LDS_BYTES = (NW * SLAB > RED_BYTES) ? NW * SLAB : RED_BYTES;
    static_assert(KG == 1 || KG == 2 || KG == 4 || KG == 8, "K split");
    static_assert(KS % D == 0, "ring depth must divide the steps per stage");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int L, int NF, int KG, int DEPTH>
__global__ __launch_bounds__(512, 2) void gemm_i8_rows32_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, float *__restrict__ out,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles, int m_slots)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Rows32Cfg<L, NF, KG, DEPTH>;
    constexpr int KS = C::KS, D = C::D, NG = C::NG;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NG, ng = wave - kg * NG;            // this wave's K group and column group
    const int l31 = lane & 31, g = lane >> 5;

    int n_real = m_slots * n_tiles;
    if (tpe != nullptr) {
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) (void)expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
        const int m_tiles = ct < m_slots ? ct : m_slots;
        n_real = m_tiles * n_tiles;
    }
    n_real = __builtin_amdgcn_readfirstlane(n_real);

  for (int vb = blockIdx.x; vb < n_real; vb += gridDim.x) {
    // ---- tile -> (expert, 32-row block, column block); m-tile major so neighbours share activations in L2
    int e = 0, row0 = 0, rows_valid = 0, prow0 = 0;
    const int tile = xcd_remap(vb, n_real);
    const int ms = tile / n_tiles;
    const int nt = tile - ms * n_tiles;
    if (tpe == nullptr) {
        row0 = prow0 = ms * C::BM;
        rows_valid = T - row0;
    } else {
        int cp = 0, ct = 0;
        bool found = false;
        for (int base = 0; base < E && !found; base += 64) {
            const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
            const unsigned long long hit = __ballot(ms >= x.tile_excl && ms < x.tile_excl + x.tiles);
            if (hit) {
                const int src = __ffsll((long long)hit) - 1;
                const int lo = __shfl(x.lo, src, 64), cnt = __shfl(x.cnt, src, 64);
                const int te = __shfl(x.tile_excl, src, 64), pe = __shfl(x.pad_excl, src, 64);
                e = base + src;
                row0 = lo + (ms - te) * C::BM;
                prow0 = pe + (ms - te) * C::BM;
                rows_valid = cnt - (ms - te) * C::BM;
                found = true;
            }
        }
        if (!found) continue;                                 // uniform over the workgroup
    }
    if (rows_valid <= 0) continue;
    if (rows_valid > C::BM) rows_valid = C::BM;
    const int n0 = nt * C::BN + ng * NF * 32;                 // first column of this wave
    e = __builtin_amdgcn_readfirstlane(e);
    row0 = __builtin_amdgcn_readfirstlane(row0);
    prow0 = __builtin_amdgcn_readfirstlane(prow0);
    rows_valid = __builtin_amdgcn_readfirstlane(rows_valid);

    const int KB = Kp / FQL_KB;
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)L * KB * MBT * 8192), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);

    const int mb = prow0 >> 5;
    int aoff[L];
#pragma unroll
    for (int l = 0; l < L; ++l) aoff[l] = ((l * KB) * MBT + mb) * 8192 + lane * 16;
    const int a_stage = MBT * 8192;

    // ---- wave-private weight slab: piece i = rows 8i..8i+7 of this wave's NF*32 rows, 128 B each
    char *slab = lds + wave * C::SLAB;
    int voffB[C::PIECES], wB[C::PIECES];
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
        const int row = i * 8 + (lane >> 3), ch = lane & 7;
        voffB[i] = (n0 + row) * (K >> 1) + ch * 16;          // rows past N fall outside the descriptor: zeros
        wB[i] = row * 128 + 16 * (ch ^ ((row >> 1) & 7));
    }
    int rB[NF], swB[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = j * 32 + l31;
        rB[j] = n * 128;
        swB[j] = (n >> 1) & 7;
    }

    v16i acc[L][NF];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;

    // ---- K loop over this K group's stages kg, kg+KG, ...: no barriers, wave-local ordering only
    const int KT = KB;
    if (kg < KT) {
        v4i bst[C::PIECES];
        v4i afr[D][L];
#pragma unroll
        for (int i = 0; i < C::PIECES; ++i) bst[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], kg * (FQL_KB / 2), 0);
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int l = 0; l < L; ++l)
                afr[s][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[l], kg * a_stage + s * 1024, 0);
        for (int kt = kg; kt < KT; kt += KG) {
            // LDS operations of one wave execute in order: these writes cannot overtake the previous stage's
            // fragment reads (all consumed by its MFMAs already)
#pragma unroll
            for (int i = 0; i < C::PIECES; ++i) *reinterpret_cast<v4i *>(slab + wB[i]) = bst[i];
            // next stage of this K group (past the end: wrong rows or zeros, never used)
#pragma unroll
            for (int i = 0; i < C::PIECES; ++i)
                bst[i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], (kt + KG) * (FQL_KB / 2), 0);
            v4i braw[NF];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int v = ks >> 1, b = ks & 1;
                if (b == 0) {
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        braw[j] = *reinterpret_cast<const v4i *>(slab + rB[j] + 16 * ((2 * v + g) ^ swB[j]));
                }
                v4i bfr[NF];
#pragma unroll
                for (int j = 0; j < NF; ++j) {
                    uint32_t lo0, hi0, lo1, hi1;
                    unpack8((uint32_t)braw[j][2 * b], lo0, hi0);
                    unpack8((uint32_t)braw[j][2 * b + 1], lo1, hi1);
                    bfr[j] = v4i{(int)lo0, (int)hi0, (int)lo1, (int)hi1};
                }
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr[j], afr[ks % D][l], acc[l][j], 0, 0, 0);
                const int nks = ks + D;                      // refill the ring slot D steps ahead
#pragma unroll
                for (int l = 0; l < L; ++l)
                    afr[ks % D][l] = __builtin_amdgcn_raw_buffer_load_b128(
                        rsA, aoff[l], (kt + (nks / KS) * KG) * a_stage + (nks % KS) * 1024, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- add the KG partial accumulators through LDS, pairwise (int32: exact, order-free).  The weight slabs
    //      are dead by the first barrier.
    if (KG > 1) {
#pragma unroll
        for (int s = 1; s < KG; s <<= 1) {
            char *red = lds + ((kg / (2 * s)) * NG + ng) * C::ACC_BYTES;
            __syncthreads();                                  // slabs / previous round's partials consumed
            if ((kg & (2 * s - 1)) == s) {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *reinterpret_cast<v4i *>(red + (((l * NF + j) * 4 + q) * 64 + lane) * 16) =
                                v4i{acc[l][j][4 * q], acc[l][j][4 * q + 1], acc[l][j][4 * q + 2], acc[l][j][4 * q + 3]};
            }
            __syncthreads();
            if ((kg & (2 * s - 1)) == 0) {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const v4i p = *reinterpret_cast<const v4i *>(red + (((l * NF + j) * 4 + q) * 64 + lane) * 16);
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[l][j][4 * q + c] += p[c];
                        }
            }
        }
        __syncthreads();                                      // the next tile's slabs overwrite the partials
    }
    if (kg != 0) continue;

    // ---- epilogue (as the wide kernel): lane owns output row t, registers 4q..4q+3 are 4 consecutive columns
    if (l31 >= rows_valid) continue;
    const int t = row0 + l31;
    const float d = delta[t];
    float rs[L];
#pragma unroll
    for (int l = 0; l < L; ++l) rs[l] = (float)rowsum[(size_t)l * T + t];
    const float *sce = scales + (size_t)e * N;
    const float *zpe = zps + (size_t)e * N;
    float *orow = out + (size_t)t * N;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(sce) & 15) == 0) && ((reinterpret_cast<uintptr_t>(zpe) & 15) == 0);
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + j * 32 + 8 * q + 4 * g;
            if (n >= N) continue;
            if (vec) {
                const v4f s4 = *reinterpret_cast<const v4f *>(sce + n);
                const v4f z4 = *reinterpret_cast<const v4f *>(zpe + n);
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tot = 0.0f;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l)
                        tot = fmaf(tot, 256.0f, fmaf(-z4[c], rs[l], (float)acc[l][j][4 * q + c]));
                    o[c] = (tot * d) * s4[c];
                }
                *reinterpret_cast<v4f *>(orow + n) = v4f{o[0], o[1], o[2], o[3]};
            } else {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (n + c >= N) continue;
                    float tot = 0.0f;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l)
                        tot = fmaf(tot, 256.0f, fmaf(-zpe[n + c], rs[l], (float)acc[l][j][4 * q + c]));
                    orow[n + c] = (tot * d) * sce[n + c];
                }
            }
        }
  }
#endif
}
