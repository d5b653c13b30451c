// Grouped INT4-weight x INT8-limb-activation GEMM on the CDNA4 matrix cores
// (v_mfma_i32_32x32x32_i8), one launch for all experts.  The linear op is the 1-group case.
//
//   out[t][n] = scale[e][n] * delta[t] * sum_l 256^l * ( sum_k q[e][n][k] * a_l[t][k]  -  zp[e][n] * rowsum_l[t] )
//
// for every row t of expert e's range.  The inner integer dot products are exact (i32
// accumulation), so results do not depend on tile shape, K order or which GPU ran the row.
//
// One workgroup = 8 waves (WM x WN), tile BM = 32*WM rows x BN = 32*NF*WN columns, 1 workgroup per CU:
//   * packed weights (the HBM stream, read once): 8 rows x 128 B = 8 FULL cache lines per wave
//     instruction, global -> VGPR -> LDS one 256-k stage ahead (two LDS stages, XOR-swizzled so the
//     fragment reads are bank-conflict-free); read back with one ds_read_b128 per 32 columns x 64 k;
//     nibbles are unpacked in registers with 3 VALU ops per 8 weights (unpack8) straight into the MFMA
//     B operand.
//   * activation limbs (re-read per column tile, served by the XCD's L2): the pre-pass stores them in
//     MFMA-fragment order, so a wave's A operand for one 32-deep k-step is ONE coalesced 1 KiB
//     buffer_load_dwordx4 straight into VGPRs -- no LDS round trip, no LDS-DMA (whose ~25 B/clk/CU
//     ceiling and ~100-cycle issue cost capped the earlier LDS-staged version); a 4-step register ring
//     keeps 4 k-steps of A in flight per wave.
//   * every load is an ordinary buffer load, so hipcc's own counted s_waitcnt vmcnt(N) tracks them;
//     one workgroup barrier per 256-k stage (8 MFMA k-steps).
//
// Replaces (reference, CUDA): csrc/quantized_linear_kernel.cu:90-279 (one thread per output,
// weights re-read per batch row) and csrc/moe_int4_kernel.cu:17-136 (one <<<1,256>>> launch and two
// host syncs per expert).
#pragma once
#include "fql_common.h"

template <int L, int WM, int WN, int NF, int DEPTH, int BDEPTH = 1>
struct GemmCfg {
    static constexpr int BD = BDEPTH;                        // weight stages in flight in the staging-register ring
    static constexpr int NW = WM * WN;
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = FQL_MB * WM;
    static constexpr int BN = 32 * NF * WN;
    static constexpr int KS = FQL_KB / 32;                   // MFMA k-steps per weight stage (8)
    static constexpr int D = DEPTH;                          // A prefetch depth in k-steps (register ring)
    static constexpr int B_STAGE = BN * (FQL_KB / 2);        // bytes of packed weights per stage
    static constexpr int LDS_BYTES = 2 * B_STAGE;
    static constexpr int CPWB = BN / 8 / NW;                 // 1 KiB weight pieces per wave per stage
    static_assert(NW == 8 || NW == 4 || NW == 2, "8 waves (two per SIMD), or small 4 / 2-wave workgroups for skinny tiles");
    static_assert(KS % D == 0, "ring depth must divide the steps per stage");
    static_assert((BN / 8) % NW == 0, "weight pieces must divide evenly over the waves");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

// Debug builds (-DFQL_TRACE, tools/trace_kernel.py): wave 0 of the first 8 workgroups stamps the shader clock
// (s_memtime) at every phase boundary and the constant 100 MHz clock (s_memrealtime) at tile boundaries.
#if defined(FQL_TRACE)
__device__ unsigned long long fql_trace_wide[8 * 64];
#define FQL_WSTAMP(i, real) do { if (blockIdx.x < 8 && threadIdx.x == 0 && (i) < 64) fql_trace_wide[blockIdx.x * 64 + (i)] = (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FQL_WSTAMP(i, real) do { } while (0)
#endif

__device__ __forceinline__ void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14)); }


template <int L, int WM, int WN, int NF, int DEPTH, int BDEPTH>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_i8_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, float *__restrict__ out,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles, int m_slots)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub
    using C = GemmCfg<L, WM, WN, NF, DEPTH, BDEPTH>;
    constexpr int KS = C::KS, D = C::D;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    // ---- tiles.  Real m-tiles are counted on the device (expert counts live there).  The launch is
    //      PERSISTENT: one workgroup per CU walks virtual block ids vb = blockIdx.x, +gridDim.x, ...; the
    //      16-byte stores of one tile's epilogue drain while the next tile's loads and MFMAs start.
    //      Logical tile ids are m-tile major and dealt to XCDs in contiguous ranges (vb % 8 = the XCD
    //      group of the workgroup, for every vb it visits), so the workgroups of one XCD share an
    //      expert's activation panel in that XCD's L2 while each weight byte streams once.
    const int lane0 = threadIdx.x & 63;
    int n_real = m_slots * n_tiles;
    if (tpe != nullptr) {
        // One vector load per 64 experts (every wave does it redundantly; nothing is shared).
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) (void)expert_chunk(tpe, offs, E, T, C::BM, base, lane0, cp, ct);
        const int m_tiles = ct < m_slots ? ct : m_slots;     // overlapping ranges: stay inside the plan
        n_real = m_tiles * n_tiles;
    }

    n_real = __builtin_amdgcn_readfirstlane(n_real);
  int ev = 0;
  for (int vb = blockIdx.x; vb < n_real; vb += gridDim.x) {
    FQL_WSTAMP(ev++, 1);                                     // tile start, constant 100 MHz clock
    FQL_WSTAMP(ev++, 0);                                     // tile start, shader clock
    int e = 0, row0 = 0, rows_valid = 0, prow0 = 0;
    const int tile = xcd_remap(vb, n_real);
    const int ms = tile / n_tiles;
    const int nt = tile - ms * n_tiles;
    if (tpe == nullptr) {                                   // linear: one group covering all T rows
        row0 = prow0 = ms * C::BM;
        rows_valid = T - row0;
    } else {                                                // MoE: the expert that owns this m-tile
        int cp = 0, ct = 0;
        bool found = false;
        for (int base = 0; base < E && !found; base += 64) {
            const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane0, cp, ct);   // re-read per tile: keeps no table live in registers
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
        if (!found) continue;
    }
    if (rows_valid <= 0) continue;
    if (rows_valid > C::BM) rows_valid = C::BM;
    const int n0 = nt * C::BN;
    e = __builtin_amdgcn_readfirstlane(e);
    row0 = __builtin_amdgcn_readfirstlane(row0);
    prow0 = __builtin_amdgcn_readfirstlane(prow0);
    rows_valid = __builtin_amdgcn_readfirstlane(rows_valid);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, g = lane >> 5;
    const bool active = wm * FQL_MB < rows_valid;           // waves past the expert's last row only help stage weights

    // ---- buffer descriptors (bounds-checked: weight rows past N and the K tail read as zero)
    const int KB = Kp / FQL_KB;
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)L * KB * MBT * 8192), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);

    // ---- A operand: limbs[l][kb][mb][ks][lane][16 B]; this wave's row block is mb
    const int mb = (prow0 >> 5) + wm;
    int aoff[L];
#pragma unroll
    for (int l = 0; l < L; ++l) aoff[l] = ((l * KB) * MBT + mb) * 8192 + lane * 16;
    const int a_stage = MBT * 8192;                         // bytes between consecutive kb of one limb

    // ---- weight staging: piece p = i*8 + wave covers rows 8p..8p+7, 128 B each (8 full lines)
    int voffB[C::CPWB], wB[C::CPWB];
#pragma unroll
    for (int i = 0; i < C::CPWB; ++i) {
        const int row = (i * C::NW + wave) * 8 + (lane >> 3), ch = lane & 7;
        voffB[i] = (n0 + row) * (K >> 1) + ch * 16;
        wB[i] = row * 128 + 16 * (ch ^ ((row >> 1) & 7));   // swizzled LDS image
    }
    int rB[NF], swB[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = (wn * NF + j) * 32 + l31;
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

    const int KT = KB;                                       // weight stages (K padded to 256 by the pre-pass)
    v4i afr[D][L];                                           // A ring: D k-steps ahead

    // ---- prologue: stage 0 of the weights into LDS, stages 1..BD into the staging-register ring (slot of
    //      stage s = s % BD), A for steps 0..D-1.  Every prefetch below is UNCONDITIONAL: past the last stage
    //      the buffer offsets fall outside the descriptors and the loads return zero.  (A load under an `if`
    //      makes hipcc's counted vmcnt collapse to "wait for almost everything", which throws the prefetch
    //      lead away.)
    constexpr int BD = C::BD;
    v4i bst[BD][C::CPWB];                                    // weight stages in flight (global -> VGPR -> LDS)
#pragma unroll
    for (int i = 0; i < C::CPWB; ++i) bst[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], 0, 0);
#pragma unroll
    for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(lds + wB[i]) = bst[0][i];
#pragma unroll
    for (int s = 1; s <= BD; ++s)
#pragma unroll
        for (int i = 0; i < C::CPWB; ++i)
            bst[s % BD][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], s * (FQL_KB / 2), 0);

    if (active) {
        // ---- the weight fragments are software-pipelined one k-step ahead: the ds_reads of the next 64-k
        //      pair and the nibble unpack of the next step are issued under the current step's MFMAs.  One
        //      barrier per stage, placed at step 5: by then every wave has parked stage kt+1 (its step 0) and
        //      has finished reading stage kt (the last read of it is issued at step 4).
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int l = 0; l < L; ++l) afr[s][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff[l], s * 1024, 0);
        wait_lgkmcnt0();
        __builtin_amdgcn_s_barrier();
        // ping-pong register sets indexed by compile-time parity (the k-step loop is fully unrolled), so the
        // hand-over from "next" to "current" costs no register moves
        v4i bfr2[2][NF], braw2[2][NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            braw2[0][j] = *reinterpret_cast<const v4i *>(lds + rB[j] + 16 * ((0 + g) ^ swB[j]));
            uint32_t lo0, hi0, lo1, hi1;
            unpack8((uint32_t)braw2[0][j][0], lo0, hi0);
            unpack8((uint32_t)braw2[0][j][1], lo1, hi1);
            bfr2[0][j][0] = (int)lo0; bfr2[0][j][1] = (int)hi0; bfr2[0][j][2] = (int)lo1; bfr2[0][j][3] = (int)hi1;
        }
        for (int kt0 = 0; kt0 < KT; kt0 += BD) {
#pragma unroll
          for (int kk = 0; kk < BD; ++kk) {                  // unrolled so the staging-ring slot is static
            const int kt = kt0 + kk;
            if (kt >= KT) break;
            const char *sb = lds + (kt & 1) * C::B_STAGE;
            char *nb = lds + ((kt + 1) & 1) * C::B_STAGE;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int v = ks >> 1, b = ks & 1;
                const int pc = v & 1, pn = (v + 1) & 1;          // raw-register set of this pair / the next pair
                if (ks == 0) {
                    FQL_WSTAMP(ev++, 0);                     // stage start
                    // the other LDS stage was released by the barrier of stage kt-1: park stage kt+1 there now,
                    // then refill that ring slot with stage kt+1+BD (BD stages of HBM lead).
                #pragma unroll
                    for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(nb + wB[i]) = bst[(kk + 1) % BD][i];
#pragma unroll
                    for (int i = 0; i < C::CPWB; ++i)
                        bst[(kk + 1) % BD][i] =
                            __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], (kt + 1 + BD) * (FQL_KB / 2), 0);
                }
                if (ks == 5) {
                    wait_lgkmcnt0();
                    __builtin_amdgcn_s_barrier();
                }
                if (b == 0) {                  // raw weights of the NEXT pair (pair 0 of the next stage at step 6)
                    const char *src = (v < 3) ? sb : nb;
                    const int nv = (v + 1) & 3;
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        braw2[pn][j] = *reinterpret_cast<const v4i *>(src + rB[j] + 16 * ((2 * nv + g) ^ swB[j]));
                }
#if defined(FQL_ABLATE) && FQL_ABLATE == 1
#pragma unroll
                for (int l = 0; l < L; ++l) asm volatile("" ::"v"(afr[ks % D][l]));
#pragma unroll
                for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(bfr2[b][j]));
#else
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr2[b][j], afr[ks % D][l], acc[l][j], 0, 0, 0);
#endif
#pragma unroll
                for (int j = 0; j < NF; ++j) {  // unpack for the next step under the MFMAs
                    uint32_t lo0, hi0, lo1, hi1;
                    if (b == 0) {
                        unpack8((uint32_t)braw2[pc][j][2], lo0, hi0);
                        unpack8((uint32_t)braw2[pc][j][3], lo1, hi1);
                    } else {
                        unpack8((uint32_t)braw2[pn][j][0], lo0, hi0);
                        unpack8((uint32_t)braw2[pn][j][1], lo1, hi1);
                    }
                    bfr2[b ^ 1][j][0] = (int)lo0; bfr2[b ^ 1][j][1] = (int)hi0;
                    bfr2[b ^ 1][j][2] = (int)lo1; bfr2[b ^ 1][j][3] = (int)hi1;
                }
                // refill the ring slot just consumed with the A fragments D steps ahead
                const int nks = ks + D;
#if !(defined(FQL_ABLATE) && FQL_ABLATE == 2)      // ablation 2: no A refills (timing experiment only, wrong results)
#pragma unroll
                for (int l = 0; l < L; ++l)
                    afr[ks % D][l] = __builtin_amdgcn_raw_buffer_load_b128(
                        rsA, aoff[l], (kt + nks / KS) * a_stage + (nks % KS) * 1024, 0);
#else
                (void)nks;
#endif
                // pin the software pipeline: without this the machine scheduler sinks the prefetch loads
                // down to their use D steps later (load; s_waitcnt vmcnt(0); mfma) to save registers.
                __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
        wait_lgkmcnt0();
    } else {
        // waves past the expert's last row: only help stage the weights and keep the barriers in step
        wait_lgkmcnt0();
        __builtin_amdgcn_s_barrier();
        for (int kt0 = 0; kt0 < KT; kt0 += BD) {
#pragma unroll
          for (int kk = 0; kk < BD; ++kk) {
            const int kt = kt0 + kk;
            if (kt >= KT) break;
            char *nb = lds + ((kt + 1) & 1) * C::B_STAGE;
#pragma unroll
            for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(nb + wB[i]) = bst[(kk + 1) % BD][i];
#pragma unroll
            for (int i = 0; i < C::CPWB; ++i)
                bst[(kk + 1) % BD][i] =
                    __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB[i], (kt + 1 + BD) * (FQL_KB / 2), 0);
            wait_lgkmcnt0();
            __builtin_amdgcn_s_barrier();
          }
        }
        continue;
    }

    FQL_WSTAMP(ev++, 0);                                     // K loop done
    // ---- epilogue: fold zero-point, combine limbs, scale.  The weights are the MFMA's A operand (rows = n)
    //      and the activations its B operand (cols = t), so in the 32x32 C/D layout
    //      (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) every lane owns ONE output
    //      row t and registers 4q..4q+3 are 4 consecutive output columns: 4 per-row loads per lane and
    //      16-byte stores.
    const int rl = wm * FQL_MB + l31;
    if (rl >= rows_valid) continue;
    const int t = row0 + rl;
    const float d = delta[t];
    float rs[L];
#pragma unroll
    for (int l = 0; l < L; ++l) rs[l] = (float)rowsum[(size_t)l * T + t];
    const float *sce = scales + (size_t)e * N;
    const float *zpe = zps + (size_t)e * N;
    float *orow = out + (size_t)t * N;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(sce) & 15) == 0) && ((reinterpret_cast<uintptr_t>(zpe) & 15) == 0);
    if (vec) {
        // scale / zero-point vectors through bounds-checked buffer loads (columns past N read as zero), issued one
        // fragment AHEAD of the arithmetic that uses them: unconditional, so they batch instead of paying one
        // memory round trip per 4 columns
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)sce, 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void *)zpe, 0, N * 4, 0x00020000);
        v4f s4[2][4], z4[2][4];
        auto fetch = [&](int j, int slot) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + (wn * NF + j) * 32 + 8 * q + 4 * g;
                s4[slot][q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsS, n * 4, 0, 0));
                z4[slot][q] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsZ, n * 4, 0, 0));
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j + 1 < NF) fetch(j + 1, (j + 1) & 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + (wn * NF + j) * 32 + 8 * q + 4 * g;
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tot = 0.0f;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l)
                        tot = fmaf(tot, 256.0f, fmaf(-z4[j & 1][q][c], rs[l], (float)acc[l][j][4 * q + c]));
                    o[c] = (tot * d) * s4[j & 1][q][c];
                }
                if (n < N) *reinterpret_cast<v4f *>(orow + n) = v4f{o[0], o[1], o[2], o[3]};   // N % 4 == 0
            }
        }
    } else {
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + (wn * NF + j) * 32 + 8 * q + 4 * g;
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
    FQL_WSTAMP(ev++, 0);                                     // epilogue issued
    FQL_WSTAMP(ev++, 1);
  }   // persistent tile loop
#endif  // __HIP_DEVICE_COMPILE__
}
