// Grouped INT4-weight x INT8-limb-activation GEMM, ONE WAVE PER SIMD (round 3).  Same arithmetic, same tiles and
// bit-for-bit the same results as gemm_i8_kernel (fql_gemm_i8.h); what differs is who does what inside a tile:
//
//   * 4 waves (one per SIMD, 512 registers each) instead of 8.  A wave owns ONE 32-row block and ALL NF column
//     fragments of the 128 x 32 NF tile (3 limbs x 6 fragments = 288 accumulator registers; the accumulators live in
//     AGPRs and VGPRs alike: this translation unit is built with -mllvm -amdgpu-mfma-vgpr-form).  So
//       - every activation fragment is loaded by exactly one wave (half the vector-memory traffic of the 4 x 2 wave
//         grid, where both column halves fetched the same rows) and feeds 6 matrix instructions instead of 3;
//       - the freed registers buy a FULL-STAGE activation ring: every load of a stage is issued one whole 256-k
//         stage (8 k-steps) before its use -- activations, weights alike;
//   * the 4-bit weights are unpacked ONCE, on their way into LDS (global -> VGPR -> unpack8 -> 2 x ds_write_b128),
//     and read back as ready matrix operands (one ds_read_b128 per fragment and k-step): 72 VALU unpack instructions
//     per wave and stage instead of 288, none of them in the matrix loop;
//   * the pipeline never drains between tiles: the weight stages and the activation ring run across the tile
//     boundary (the next tile's first stage is in LDS, its second one in flight and its first 8 k-steps of
//     activations are in the ring when the current tile's last matrix instruction issues), the per-row epilogue
//     values are loaded at the start of the tile, and the expert table / heavy-tail probe of up to 16 tiles ahead
//     is computed once per workgroup into LDS -- no vector load is ever waited for right after it is issued
//     (vector-memory operations complete in order: such a wait would drain the whole prefetch queue).
//
// LDS image of a weight stage (unpacked; per k-step a plane of NF fragment blocks, planes KSTRIDE apart; a fragment block is
// the operand of one matrix instruction in LANE ORDER: lane group g, then the fragment's 32 rows, 16 B each):
//   k-step ks, row n of the tile, lane group g  ->  byte ks * KSTRIDE + (n >> 5) * KFRAG + g * KHALF + (n & 31) * 16
//   KHALF = 32 * 16 + 16,  KFRAG = 2 * KHALF,  KSTRIDE = NF * KFRAG + 16
// Every read of a stage is ONE per-lane base register + an immediate offset (ks * KSTRIDE + j * KFRAG) -- the first layout,
// rows of 256 B with an XOR swizzle of the 16-byte slots, needed 4 vector instructions per k-step to form the address.
// Banks: a ds_read_b128 is served 16 lanes at a time over 64 banks, and 16 consecutive lanes read 256 contiguous bytes
// (row-major planes, n * 32 + g * 16, had consecutive lanes 32 B apart: two-way conflicts, SQ_LDS_BANK_CONFLICT 5.7 M of
// 15 M LDS cycles); the parking ds_write_b128s are served 8 lanes at a time over 32 banks -- the 8 chunks c = 2 v + g' of
// one row, which go to k-steps 2 v / 2 v + 1, group g': the 16 bytes of padding per lane-group half and per plane put
// them on 2 v + g' = 8 different 16-byte bank groups.
//
// Replaces (reference, CUDA): csrc/moe_int4_kernel.cu:17-136 and csrc/quantized_linear_kernel.cu:90-279.
#pragma once
#include "fql_common.h"
#include "fql_gemm_i8.h"
#include "fql_act_quant.h"
#include "fql_w4_launch.h"

// v_mfma_i32_32x32x32_i8 with the accumulator in VGPRs (the builtin's accumulators are AGPRs in a 512-register kernel).
// Hazards the compiler cannot see are excluded by construction: two of these on one accumulator are always separated
// by >= 5 other matrix instructions, and the epilogue waits 32 cycles before its first VALU read of the results.
__device__ __forceinline__ void mfma_i8_vgpr(v16i &acc, const v4i &w, const v4i &a)
{
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}
__device__ __forceinline__ void mfma_i8_vgpr_zero(v16i &acc, const v4i &w, const v4i &a)
{
    asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=&v"(acc) : "v"(w), "v"(a));
}


template <int L, int NF, int DEPTH = 8>
struct W4Cfg {
    static constexpr int D = DEPTH;                          // activation ring depth in k-steps (8 = a full stage)
    static constexpr int NW = 4;
    static constexpr int THREADS = 256;
    static constexpr int BM = 4 * FQL_MB;                    // 128 rows: one 32-row block per wave
    static constexpr int BN = 32 * NF;
    static constexpr int KS = FQL_KB / 32;                   // 8 k-steps per stage
    static constexpr int KHALF = FQL_MB * 16 + 16;           // one lane group of a fragment block: 32 rows x 16 B (+ 16 B: bank spread of the parking stores)
    static constexpr int KFRAG = 2 * KHALF;                  // one fragment block = one matrix instruction's weight operand
    static constexpr int KSTRIDE = NF * KFRAG + 16;          // bytes between the k-step planes of a stage in LDS
    static constexpr int W_STAGE = KS * KSTRIDE;             // bytes of UNPACKED weights per stage
    static constexpr int SZ_BYTES = 2 * 3 * BN * 4;          // two scale / zero-point / bias slices
    static constexpr int NTAB = 16;                          // tiles described ahead in LDS
    static constexpr int TAB_INTS = 8;
    static constexpr int LDS_BYTES = 2 * W_STAGE + SZ_BYTES + NTAB * TAB_INTS * 4;
    static constexpr int SZN = (3 * BN + THREADS - 1) / THREADS;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static_assert(L * NF * 16 <= 320, "accumulator registers");
    static_assert((FQL_KB / 32) % DEPTH == 0, "ring depth must divide the steps per stage");
};

// Timing experiments only (wrong results), bit mask: 1 no weight park / staging loads, 2 no activation ring refills,
// 4 no weight-fragment reads, 8 no branch around the last fragment (always all NF), 16 no epilogue arithmetic / stores
#ifndef W4_SHORT_RING
#define W4_SHORT_RING 1
#endif
#ifndef W4_ABLATE
#define W4_ABLATE 0
#endif
#if defined(FQL_TRACE)
__device__ unsigned long long fql_trace_w4[8 * 64];
#define FQL_W4STAMP(i, real) do { const int i_ = (i); if (blockIdx.x < 8 && threadIdx.x == 0 && i_ < 64) fql_trace_w4[blockIdx.x * 64 + i_] = (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FQL_W4STAMP(i, real) do { } while (0)
#endif

// FUSED (round 3): ONE launch for pre-pass + GEMM.  Every workgroup first quantises its share of the grouped rows (4 at a
// time, act_rows of fql_act_quant.h -- the pre-pass kernel's own body) while its first tile's weight stage is already on
// its way from HBM, publishes each group of rows with a release store of the launch's token, and waits for the row groups
// of ITS OWN tiles only.  No workgroup ever depends on another one for good: after a bounded number of polls it quantises
// the rows it is waiting for itself (the same bytes, so the race with their owner is benign) -- a launch can therefore not
// hang however the device schedules its workgroups (two such launches on two streams, a busy device), it only gets slower.
// What it saves (tools/trace_step.py): the launch boundary between the two kernels (~3 us) and the GEMM's prologue
// (expert scan, tile table, first weight stage: ~6 us), which now run under the pre-pass.
template <int L, int NF, int DEPTH, bool FUSED = false>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, void *__restrict__ out, int out_kind_flags,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles_min, int m_slots, float *__restrict__ res_scratch,
    const float *__restrict__ bias, int n_tiles_alt, FqlW4Fused fz)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using C = W4Cfg<L, NF, DEPTH>;
    constexpr int KS = C::KS, D = C::D;
    // The compiler gives the matrix instruction's builtin AGPR accumulators (256 registers: 16 tiles of 16); the
    // fragments past that budget accumulate in VGPRs through the instruction's VGPR form (inline assembly).
    constexpr int NVF = (L * NF * 16 > 256) ? NF - 256 / (L * 16) : 0;
    constexpr bool RES = FQL_RES_ENABLED && (L >= 2);
    constexpr int OOB = 0x7fff0000;
    // out_kind_flags: bits 0-1 the output element type, bit 3: multiply every output row by its row weight (fql_gemm_i8.h)
    const int out_kind = out_kind_flags & 3;
    const bool row_scaled = (out_kind_flags & 8) != 0;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    FQL_W4STAMP(56, 1);                                      // (trace builds: kernel entry, 100 MHz clock)
    // ---- what a wave does in a tile depends on the tile's rows (short row groups: skewed routing, the tail of a group):
    //        > 64 rows: wave w owns row block w and all NF fragments                      (18 matrix instructions per k-step)
    //      33..64 rows: wave w owns row block w & 1 and fragments 3 (w >> 1) .. + 2          (9)
    //       <= 32 rows: waves 0..2 own row block 0 and fragments 2 w, 2 w + 1; wave 3 idles  (6)
    //      so a short tile costs its weight stream and staging, not a 128-row tile's matrix work.  The integer sums of an
    //      output do not depend on which wave made them: bit-identical.
    auto tile_class = [&](const GemmTile &tp) -> int { return tp.rows_valid > 64 ? 4 : (tp.rows_valid > 32 ? 2 : 1); };
    auto wave_rb = [&](int cls) -> int { return cls == 4 ? wave : (cls == 2 ? (wave & 1) : 0); };
    auto wave_fbase = [&](int cls) -> int { return cls == 4 ? 0 : (cls == 2 ? 3 * (wave >> 1) : 2 * wave); };
    auto wave_has_work = [&](int cls) -> bool { return cls != 1 || wave < 3; };

    // ---- column tiling: exactly the wide kernel's (fql_gemm_i8.h): the N / 32 fragments of a row block dealt as evenly
    //      as possible over the tile count picked here from the real row-block count
    int n_tiles = n_tiles_min;
    int n_real = 0;
    const int n_frag = (N + 31) >> 5;
    auto pick_tiles = [&](int m_tiles) {
        if (n_tiles_alt <= 0) return;
        const int G = (int)gridDim.x;
        const float ra = (float)((m_tiles * n_tiles_min + G - 1) / G), rb = (float)((m_tiles * n_tiles_alt + G - 1) / G);
        const float ca = ra * (float)(n_frag + n_tiles_min) * (float)n_tiles_alt;
        const float cb = rb * (float)(n_frag + n_tiles_alt) * (float)n_tiles_min;
        if (cb < ca) n_tiles = n_tiles_alt;
    };
    // ---- row tiles, ordered by COST: first every expert's tiles with more than 64 rows, then the 33..64-row tails, then
    //      the tails of at most 32 rows (a short tile costs about 0.6 of a full one).  A workgroup walks tile ids
    //      blockIdx, + gridDim, ..., so every workgroup gets its share of the expensive tiles first and the cheap ones fill
    //      the last round; inside each of the three groups an XCD still owns a contiguous range of tile ids (its
    //      workgroups share an expert's activation panel in that XCD's L2).  Under even routing there is one group and
    //      the order is the wide kernel's.
    struct RowGroups { int lo, cnt, pad_excl, nbig, big_excl, c2, c2_excl, c1, c1_excl; };
    auto row_groups = [&](int base, int &carry_pad, int &cb, int &c2, int &c1) -> RowGroups {
        RowGroups r;
        r.lo = 0; r.cnt = 0;
        if (base + lane < E) expert_range(tpe, offs, base + lane, T, r.lo, r.cnt);
        const int pad = (r.cnt + FQL_MB - 1) / FQL_MB * FQL_MB;
        const int full = r.cnt / C::BM, rem = r.cnt - full * C::BM;
        r.nbig = full + (rem > 64 ? 1 : 0);
        r.c2 = (rem > 32 && rem <= 64) ? 1 : 0;
        r.c1 = (rem >= 1 && rem <= 32) ? 1 : 0;
        const int pad_incl = wave_incl_scan(pad, lane), big_incl = wave_incl_scan(r.nbig, lane);
        const int c2_incl = wave_incl_scan(r.c2, lane), c1_incl = wave_incl_scan(r.c1, lane);
        r.pad_excl = carry_pad + pad_incl - pad;
        r.big_excl = cb + big_incl - r.nbig;
        r.c2_excl = c2 + c2_incl - r.c2;
        r.c1_excl = c1 + c1_incl - r.c1;
        carry_pad += wave_bcast(pad_incl, 63);
        cb += wave_bcast(big_incl, 63);
        c2 += wave_bcast(c2_incl, 63);
        c1 += wave_bcast(c1_incl, 63);
        return r;
    };
    int MB = m_slots, MC2 = 0, MC1 = 0;                      // row tiles per cost group
    // (the first 64 experts' scan is kept for the prologue's tile table: one load round trip and four wave scans less on
    //  the way to the first weight load; later table refills scan again, so nothing of it stays live in the matrix loop)
    RowGroups rg0 = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int cp0 = 0, cb0 = 0, c20 = 0, c10 = 0;
    if (tpe != nullptr) {
        int cp = 0, cb = 0, c2 = 0, c1 = 0;
        rg0 = row_groups(0, cp, cb, c2, c1);
        cp0 = cp; cb0 = cb; c20 = c2; c10 = c1;
        for (int base = 64; base < E; base += 64) (void)row_groups(base, cp, cb, c2, c1);
        MB = __builtin_amdgcn_readfirstlane(cb); MC2 = __builtin_amdgcn_readfirstlane(c2); MC1 = __builtin_amdgcn_readfirstlane(c1);
        if (MB + MC2 + MC1 > m_slots) {                      // overlapping ranges: stay inside the plan
            MB = MB < m_slots ? MB : m_slots;
            MC2 = MC2 < m_slots - MB ? MC2 : m_slots - MB;
            MC1 = MC1 < m_slots - MB - MC2 ? MC1 : m_slots - MB - MC2;
        }
        if (MC2 + MC1 == 0) pick_tiles(MB);
        else if (n_tiles_alt > 0) {
            // mixed costs: replay the walk for both tile counts -- the expensive tiles fill whole rounds, the cheap ones
            // (0.6 of a full tile) continue where those end -- and take the one whose busiest workgroup carries less
            auto busiest = [&](int t) -> float {
                const int G = (int)gridDim.x;
                const int nb = MB * t, ns = (MC2 + MC1) * t;
                const int R = nb % G;                        // workgroups with one expensive tile more
                const int sq = ns / G, sr = ns - sq * G;     // cheap tiles: sq each, one more for sr workgroups from R on
                const float cb = (float)n_frag / (float)t + 1.0f, cs = 0.6f * cb;
                const float hi = (float)(nb / G + (R > 0 ? 1 : 0)), lo = (float)(nb / G);
                // workgroups [0, R): hi expensive; [R, G): lo expensive; extra cheap tile for [R, R + sr) wrapping to [0, R + sr - G)
                const int wrap = R + sr - G;                 // > 0: the extra cheap tiles reach the workgroups with hi
                float m = lo * cb + (float)(sq + (sr > 0 ? 1 : 0)) * cs;
                const float mh = hi * cb + (float)(sq + (wrap > 0 ? 1 : 0)) * cs;
                if (R > 0 && mh > m) m = mh;
                return m;
            };
            if (busiest(n_tiles_alt) < busiest(n_tiles_min)) n_tiles = n_tiles_alt;
        }
    } else {
        pick_tiles(m_slots);
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    const int n_big = MB * n_tiles, n_c2 = MC2 * n_tiles, n_c1 = MC1 * n_tiles;
    n_real = __builtin_amdgcn_readfirstlane(n_big + n_c2 + n_c1);
    const bool has_tiles = (int)blockIdx.x < n_real;
    if (!FUSED && !has_tiles) return;
    FQL_W4STAMP(57, 1);                                      // (row groups counted)
    const int f_base = n_frag / n_tiles, f_rem = n_frag - f_base * n_tiles;

    // slots [lo, lo + cnt) of the walk order -> tile ids [lo, lo + cnt), the slots of one XCD (slot & 7: blocks b and b + 8
    // share an XCD, and gridDim is a multiple of 8) getting a contiguous range.  Speed only, never correctness.
    auto xcd_range_remap = [&](int vb, int lo, int cnt) -> int {
        auto upto = [&](int v, int x) -> int { return v <= x ? 0 : ((v - 1 - x) >> 3) + 1; };   // slots < v with slot & 7 == x
        const int x = vb & 7;
        int basex = 0;
        for (int xx = 0; xx < 8; ++xx)
            if (xx < x) basex += upto(lo + cnt, xx) - upto(lo, xx);
        return lo + basex + (upto(vb, x) - upto(lo, x));
    };

    auto tile_params = [&](int vb, auto cached_tag) -> GemmTile {
        constexpr bool CACHED = decltype(cached_tag)::value;
        GemmTile tp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (vb >= n_real) return tp;
        const int grp = vb < n_big ? 0 : (vb < n_big + n_c2 ? 1 : 2);
        const int glo = grp == 0 ? 0 : (grp == 1 ? n_big : n_big + n_c2);
        const int tile = xcd_range_remap(vb, glo, grp == 0 ? n_big : (grp == 1 ? n_c2 : n_c1)) - glo;
        const int ms = tile / n_tiles;                       // row tile inside its cost group
        tp.nt = tile - ms * n_tiles;
        tp.nfr = f_base + (tp.nt < f_rem ? 1 : 0);
        tp.n0 = (tp.nt * f_base + (tp.nt < f_rem ? tp.nt : f_rem)) * 32;
        if (tpe == nullptr) {
            tp.row0 = tp.prow0 = ms * C::BM;
            tp.rows_valid = T - tp.row0;
            tp.ok = 1;
        } else {
            int cp = CACHED ? cp0 : 0, cb = CACHED ? cb0 : 0, c2 = CACHED ? c20 : 0, c1 = CACHED ? c10 : 0;
            for (int base = 0; base < E && !tp.ok; base += 64) {
                const RowGroups x = (CACHED && base == 0) ? rg0 : row_groups(base, cp, cb, c2, c1);
                const bool mine = grp == 0 ? (ms >= x.big_excl && ms < x.big_excl + x.nbig)
                                           : (grp == 1 ? (x.c2 && ms == x.c2_excl) : (x.c1 && ms == x.c1_excl));
                const unsigned long long hit = __ballot(mine);
                if (hit) {
                    const int src = __ffsll((long long)hit) - 1;
                    const int lo = wave_bcast(x.lo, src), cnt = wave_bcast(x.cnt, src);
                    const int be = wave_bcast(x.big_excl, src), pe = wave_bcast(x.pad_excl, src);
                    const int slice = grp == 0 ? ms - be : cnt / C::BM;       // the tail is the slice after the full ones
                    tp.e = base + src;
                    tp.row0 = lo + slice * C::BM;
                    tp.prow0 = pe + slice * C::BM;
                    tp.rows_valid = cnt - slice * C::BM;
                    tp.ok = 1;
                }
            }
        }
        if (tp.rows_valid <= 0) tp.ok = 0;
        if (tp.rows_valid > C::BM) tp.rows_valid = C::BM;
        return tp;
    };

    // ---- the tiles this workgroup will visit, NTAB at a time, described in LDS: {e, row0, prow0, rows_valid, n0, nfr,
    //      ok, has-heavy-tailed-rows}.  Wave w fills entries w, w + 4, ...; later reads are LDS broadcasts.
    int *tab = reinterpret_cast<int *>(lds + 2 * C::W_STAGE + C::SZ_BYTES);
    // (probe = false: the heavy-tail probe of the entries is left to probe_table -- the kernel prologue issues the first
    //  tile's weight loads between the two, so that the probe's round trip to memory runs under theirs)
    auto fill_table = [&](int first, auto in_loop_tag) {
        constexpr bool probe = decltype(in_loop_tag)::value;   // (the prologue's call: cached scan, probe deferred)
        for (int i = wave; i < C::NTAB; i += C::NW) {
            const long long vbl = (long long)blockIdx.x + (long long)(first + i) * (long long)gridDim.x;
            GemmTile tp = tile_params(vbl < (long long)n_real ? (int)vbl : n_real, std::integral_constant<bool, !probe>{});
            int hr = 0;
            if constexpr (RES) {
                if (probe && res_scratch != nullptr && tp.ok) hr = tile_has_residual(delta, T, tp, C::BM, lane);
            }
            if (lane == 0) {
                int *p = tab + i * C::TAB_INTS;
                p[0] = tp.e; p[1] = tp.row0; p[2] = tp.prow0; p[3] = tp.rows_valid;
                p[4] = tp.n0; p[5] = tp.nfr; p[6] = tp.ok | (vbl < (long long)n_real ? 2 : 0); p[7] = hr;
            }
        }
        __syncthreads();
    };
    // the probe in two halves: `issue` its loads for this wave's entries, `finish` evaluates them and completes the table.
    // (vector loads complete in order: issued BEFORE the first weight loads, the probe waits for nothing but itself)
    constexpr int NPR = C::NTAB / C::NW;
    ResidualProbe prb[NPR];
    auto probe_issue = [&]() {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            const int *p = tab + (wave + j * C::NW) * C::TAB_INTS;
            GemmTile tp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            tp.row0 = __builtin_amdgcn_readfirstlane(p[1]);
            tp.rows_valid = __builtin_amdgcn_readfirstlane(p[3]);
            tp.ok = __builtin_amdgcn_readfirstlane(p[6]) & 1;
            prb[j] = residual_probe_issue(delta, T, tp, C::BM, lane, RES && res_scratch != nullptr);
        }
    };
    auto probe_finish = [&]() {
#pragma unroll
        for (int j = 0; j < NPR; ++j) {
            const int hr = residual_probe_eval(prb[j]);
            if (lane == 0) tab[(wave + j * C::NW) * C::TAB_INTS + 7] = hr;
        }
        __syncthreads();
    };
    auto load_tile = [&](int slot) -> GemmTile {
        const v4i a = *reinterpret_cast<const v4i *>(tab + slot * C::TAB_INTS);
        const v4i b = *reinterpret_cast<const v4i *>(tab + slot * C::TAB_INTS + 4);
        GemmTile tp;
        tp.e = __builtin_amdgcn_readfirstlane(a[0]);
        tp.row0 = __builtin_amdgcn_readfirstlane(a[1]);
        tp.prow0 = __builtin_amdgcn_readfirstlane(a[2]);
        tp.rows_valid = __builtin_amdgcn_readfirstlane(a[3]);
        tp.n0 = __builtin_amdgcn_readfirstlane(b[0]);
        tp.nfr = __builtin_amdgcn_readfirstlane(b[1]);
        const int okbits = __builtin_amdgcn_readfirstlane(b[2]);
        tp.ok = okbits & 1;
        tp.nt = okbits >> 1;                                 // (here: 1 while the workgroup's tile list goes on)
        tp.rp = __builtin_amdgcn_readfirstlane(b[3]);
        tp.ad = 0;
        return tp;
    };

    // ---- per-lane constants
    const int KT = Kp / FQL_KB;                              // weight stages per tile (>= 2: the host sends shorter K elsewhere)
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const int a_stage = MBT * 8192;                          // bytes between consecutive 256-k blocks of one limb
    const int a_limb = KT * MBT * 8192;                      // bytes between the limbs
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)(RES ? 2 : 1) * L * KT * MBT * 8192), 0x00020000);
    const int aoff0 = lane * 16;
    // weight staging: piece i of a stage = rows 32 i + 8 wave + (lane >> 3), 16-byte chunk c = lane & 7 of the row's
    // 128-byte stage segment; chunk c = 2v + g' unpacks to k-step 2v (slot c) and k-step 2v + 1 (slot 8 + c), lane group g'
    const int rowW = wave * 8 + (lane >> 3), chW = lane & 7;
    const int voffW = rowW * (K >> 1) + chW * 16;
    const int pieceW = 32 * (K >> 1);
    const int wA0 = (chW & ~1) * C::KSTRIDE + (chW & 1) * C::KHALF + rowW * 16;   // k-step 2 v, group g', row   (+ i * KFRAG: piece i)
    const int wA1 = wA0 + C::KSTRIDE;                                             // k-step 2 v + 1
    // fragment reads: k-step ks, row n = 32 j + l31, group g  ->  ks * KSTRIDE + j * KFRAG + (g * KHALF + l31 * 16)
    const int rF0 = g * C::KHALF + l31 * 16;
    float *szbuf = reinterpret_cast<float *>(lds + 2 * C::W_STAGE);

    v4i bst[NF];                                             // one packed weight stage in flight (global -> VGPR)
    // activation ring.  Tiles with more than 64 rows (288 accumulator registers) keep D k-steps in flight in slots
    // 0 .. D-1; the short tile classes have the registers for a FULL stage (slot = k-step, 7 steps ahead): vector loads
    // complete in order, so a weight piece from HBM has to land before the activation fragment issued after it is used --
    // D - 1 k-steps later.  On a short tile a k-step is 6 or 9 matrix instructions: 3 of them (~0.3 us) are less than the
    // memory latency, and every stage stalled for the difference.  Between visits the contract stays "k-steps 0 .. D-2 of
    // the first stage in slots 0 .. D-2": a short visit ramps up in its first stage and down in its last.
    v4i afr[D][L];
    v4i wf[NF];                                              // weight fragments of the coming k-step
    float szr[C::SZN];

    auto weight_rsrc = [&](int e) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);
    };
    // scalar offsets of a tile's operands (OOB: the loads read zero and move nothing)
    auto w_base = [&](const GemmTile &tp) -> int { return tp.ok ? tp.n0 * (K >> 1) : OOB; };
    auto a_base = [&](const GemmTile &tp) -> int {
        const int cls = tile_class(tp), rb = wave_rb(cls);
        const bool act = tp.ok && wave_has_work(cls) && rb * FQL_MB < tp.rows_valid;
        return act ? ((tp.prow0 >> 5) + rb) * 8192 + ((RES && tp.rp) ? L * a_limb : 0) : OOB;
    };
    auto issue_weights = [&](const __amdgpu_buffer_rsrc_t rs, int sW, int nfr) {
#pragma unroll
        for (int i = 0; i < NF; ++i)
            bst[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voffW, (sW != OOB && i < nfr) ? sW + i * pieceW : OOB, 0);
    };
    auto issue_weight_piece = [&](const __amdgpu_buffer_rsrc_t rs, int sW, int nfr, int i) {
        bst[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voffW, (sW != OOB && i < nfr) ? sW + i * pieceW : OOB, 0);
    };
    auto park_piece = [&](char *buf, int i) {                // unpack one packed piece into its two operand slots
        uint32_t lo0, hi0, lo1, hi1, lo2, hi2, lo3, hi3;
        unpack8((uint32_t)bst[i][0], lo0, hi0);
        unpack8((uint32_t)bst[i][1], lo1, hi1);
        unpack8((uint32_t)bst[i][2], lo2, hi2);
        unpack8((uint32_t)bst[i][3], lo3, hi3);
        *reinterpret_cast<v4i *>(buf + wA0 + i * C::KFRAG) = v4i{(int)lo0, (int)hi0, (int)lo1, (int)hi1};
        *reinterpret_cast<v4i *>(buf + wA1 + i * C::KFRAG) = v4i{(int)lo2, (int)hi2, (int)lo3, (int)hi3};
    };
    auto issue_sz = [&](const GemmTile &tp) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc((void *)(scales + (size_t)tp.e * N), 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc((void *)(zps + (size_t)tp.e * N), 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(bias != nullptr ? bias + (size_t)tp.e * N : scales), 0, bias != nullptr ? N * 4 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::SZN; ++i) {
            const int idx = tid + i * C::THREADS;
            const int arr = idx < C::BN ? 0 : (idx < 2 * C::BN ? 1 : 2);
            const int col = idx - arr * C::BN;
            const int so = (tp.ok && idx < 3 * C::BN) ? 0 : OOB;
            const int vo = (tp.n0 + col) * 4;
            const int vs = __builtin_amdgcn_raw_buffer_load_b32(rsS, arr == 0 ? vo : OOB, so, 0);
            const int vz = __builtin_amdgcn_raw_buffer_load_b32(rsZ, arr == 1 ? vo : OOB, so, 0);
            const int vb = __builtin_amdgcn_raw_buffer_load_b32(rsBi, arr == 2 ? vo : OOB, so, 0);
            szr[i] = __builtin_bit_cast(float, vs | vz | vb);
        }
    };
    // (the zero points are parked NEGATED: the epilogue's fmaf(-zp, rowsum, acc) then needs no sign flip per use)
    auto park_sz = [&](float *sz) {
#pragma unroll
        for (int i = 0; i < C::SZN; ++i) {
            const int idx = tid + i * C::THREADS;
            if (idx < 3 * C::BN) sz[idx] = (idx >= C::BN && idx < 2 * C::BN) ? -szr[i] : szr[i];
        }
    };
    // the fragments a visit expects in wf[] when its first k-step starts: the first four of its wave's range, k-step 0 of
    // the stage in `buf` (every slot plan reads its later ones itself).  The plans prefetch the coming step's fragments
    // with the CURRENT tile's fragment range, so at a visit boundary they are read again for the next tile's.
    auto read_first_frags = [&](const char *buf, int fb) {
        const char *p = buf + fb * C::KFRAG + rF0;
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[j] = *reinterpret_cast<const v4i *>(p + j * C::KFRAG);
    };

    // ---- kernel prologue: the state every visit starts from
    //        LDS buffer fs & 1 holds stage 0 of the visit, bst its stage 1 (in flight), afr the A fragments of its
    //        stage 0 (in flight), wf the weight fragments of its k-step 0, sz[parity] its scale / zero-point slice
    // ---- FUSED: the pre-pass as this kernel's first phase (see the comment at the kernel's head)
    __shared__ int s_fused_missing;
    // (the by-value argument struct taken apart here: captured whole by the lambdas below it is copied to scratch memory)
    const void *const fz_x = fz.x;
    const int32_t *const fz_gather = fz.gather;
    const int fz_n_src = fz.n_src, fz_spin_limit = fz.spin_limit;
    const float *const fz_row_weight = fz.row_weight;
    unsigned long long *const fz_flags = fz.flags;
    const unsigned long long fz_token = fz.token;
    // (always_inline: an out-of-line lambda takes its captures by address, which puts T, K, the pointers ... into scratch memory)
    auto fused_rows = [&](int grp) __attribute__((always_inline)) {   // one group of 4 grouped rows, end to end
        act_rows<L, true, 0, false, false, 4, true>(fz_x, fz_gather, fz_n_src, const_cast<float *>(delta), const_cast<int32_t *>(rowsum),
                                              const_cast<int8_t *>(limbs), T, K, KT, MBT, 0, tpe, offs, E, fz_row_weight, grp * 4);
    };
    auto fused_phase1 = [&]() __attribute__((always_inline)) {
        const int ngroups = (T + 3) >> 2;
        for (int grp = (int)blockIdx.x; grp < ngroups; grp += (int)gridDim.x) {
            fused_rows(grp);
            // this thread's limb / delta / row-sum stores were write-through (sc1): once they have completed they are in
            // device-coherent memory; no L2 write-back (a release fence here: 10-20 us per workgroup)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(fz_flags + grp, fz_token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // rows of `out` no expert covers (reference semantics: torch::zeros, csrc/moe_int4_kernel.cu:109)
        if (tpe != nullptr)
            for (int zb = (int)blockIdx.x; zb * 256 < T; zb += (int)gridDim.x) act_zero_uncovered(zb, out, out_kind == 0 ? 4 : 2, N, tpe, offs, E, T);
    };
    // the row groups of this workgroup's own tiles: poll their flags (bounded), quantise what is still missing myself
    auto fused_wait = [&]() __attribute__((always_inline)) {
        if (tid == 0) s_fused_missing = 0;
        __syncthreads();
        bool miss = false;
        for (int j = 0; j < NPR; ++j) {
            const int *p = tab + (wave + j * C::NW) * C::TAB_INTS;
            const int row0 = __builtin_amdgcn_readfirstlane(p[1]), rows = __builtin_amdgcn_readfirstlane(p[3]);
            if (!(__builtin_amdgcn_readfirstlane(p[6]) & 1) || rows <= 0) continue;
            const int g0 = row0 >> 2, g1 = (row0 + rows - 1) >> 2;
            int spins = 0;
            for (;;) {
                bool all_ok = true;
                for (int grp = g0 + lane; grp <= g1; grp += 64)
                    all_ok = all_ok && (__hip_atomic_load(fz_flags + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == fz_token);
                if (__all(all_ok)) break;
                if (++spins > fz_spin_limit) { miss = true; break; }
                __builtin_amdgcn_s_sleep(8);
            }
        }
        if (miss && lane == 0) s_fused_missing = 1;
        __syncthreads();
        if (s_fused_missing) {                               // (never on an idle device: the owners are co-resident and as fast as we are)
            for (int i = 0; i < C::NTAB; ++i) {
                const int *p = tab + i * C::TAB_INTS;
                const int row0 = __builtin_amdgcn_readfirstlane(p[1]), rows = __builtin_amdgcn_readfirstlane(p[3]);
                if (!(__builtin_amdgcn_readfirstlane(p[6]) & 1) || rows <= 0) continue;
                for (int grp = row0 >> 2; grp <= ((row0 + rows - 1) >> 2); ++grp) {
                    __syncthreads();
                    fused_rows(grp);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows other workgroups wrote: not from this XCD's stale cache lines
        __syncthreads();
    };
    if constexpr (FUSED) {
        if (!has_tiles) { fused_phase1(); return; }          // no tile for this workgroup: its share of the rows, then done
    }

    fill_table(0, std::false_type{});
    FQL_W4STAMP(58, 1);                                      // (tile table in LDS)
    int ti = 0;                                              // tile index of this workgroup (table slot ti % NTAB)
    GemmTile cur = load_tile(0);
    int fs = 0;                                              // stages since kernel start: LDS buffer parity
    int parity = 0;
    {
        const __amdgpu_buffer_rsrc_t rs = weight_rsrc(cur.e);
        const int sW = w_base(cur);
        if constexpr (FUSED) {
            // the first weight stage leaves for HBM before the rows are quantised; the probe after the rows exist
            issue_weights(rs, sW, cur.nfr);
            issue_sz(cur);
            fused_phase1();
            FQL_W4STAMP(60, 1);                              // (own rows quantised and published)
            fused_wait();
            FQL_W4STAMP(61, 1);                              // (the tiles' rows are there)
            probe_issue();
            probe_finish();
        } else {
            // which limb set the first visit reads depends on the heavy-tail probe: its round trip runs under the weights'
            probe_issue();
            issue_weights(rs, sW, cur.nfr);
            issue_sz(cur);
            probe_finish();
        }
        cur = load_tile(0);
        if constexpr (!RES) cur.rp = 0;
        const int sA = a_base(cur);
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int l = 0; l < L; ++l)
                afr[s][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, sA == OOB ? OOB : sA + l * a_limb + s * 1024, 0);
#pragma unroll
        for (int i = 0; i < NF; ++i) park_piece(lds, i);
        park_sz(szbuf);
        issue_weights(rs, sW == OOB ? OOB : sW + (FQL_KB / 2), cur.nfr);
        __syncthreads();
        read_first_frags(lds, wave_fbase(tile_class(cur)));
    }
    FQL_W4STAMP(59, 1);                                      // (first stage parked, second in flight)

    int ev = 0; (void)ev;
    for (;;) {                                               // one iteration per VISIT (tile, pass)
        FQL_W4STAMP(ev++, 1);
        FQL_W4STAMP(ev++, 0);
        const bool rpass = RES && cur.rp != 0;
        const int cls = tile_class(cur), wm = wave_rb(cls), fbase = wave_fbase(cls);
        const bool active = cur.ok && wave_has_work(cls) && wm * FQL_MB < cur.rows_valid;
        // ---- the visit after this one: the main pass of the same tile after its residual pass, else the next tile
        GemmTile nxt;
        if (rpass) { nxt = cur; nxt.rp = 0; nxt.ad = 1; }
        else {
            ++ti;
            if ((ti % C::NTAB) == 0) fill_table(ti, std::true_type{});         // (rare: more than 16 tiles per workgroup; drains the pipeline)
            nxt = load_tile(ti % C::NTAB);
            if constexpr (!RES) nxt.rp = 0;
        }
        // ---- per-row values of this visit's epilogue and the next visit's scale / zero-point slice: issued now, used
        //      a whole K loop later
        const int rl = wm * FQL_MB + l31;
        const bool row_ok = active && rl < cur.rows_valid;
        const int t = row_ok ? cur.row0 + rl : 0;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)delta, 0, ((RES ? 2 : 1) + 1) * T * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)rowsum, 0, (RES ? 2 : 1) * L * T * 4, 0x00020000);
        const int tsel = rpass ? T : 0;
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (tsel + t) * 4, 0, 0));
        int rsi[L];
#pragma unroll
        for (int l = 0; l < L; ++l) rsi[l] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (L * tsel + l * T + t) * 4, 0, 0);
        const int d2bits = RES ? __builtin_amdgcn_raw_buffer_load_b32(rsD, (T + t) * 4, 0, 0) : 0;
        const bool addp = RES && (d2bits & 0x7fffffff) != 0;
        const float rw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, row_scaled ? ((RES ? 2 : 1) * T + t) * 4 : OOB, 0, 0));
        issue_sz(nxt);

        const __amdgpu_buffer_rsrc_t rsWc = weight_rsrc(cur.e), rsWn = weight_rsrc(nxt.e);
        const int sWc = w_base(cur), sWn = w_base(nxt);
        const int sAc = a_base(cur), sAn = a_base(nxt);
        int sA0 = sAc;                                       // activation offset of the stage being computed

        using T_ = std::true_type;
        using F_ = std::false_type;
        auto visit_body = [&](auto nact_tag) {
        constexpr int NACT = decltype(nact_tag)::value;      // fragments a wave holds in this kind of tile
        constexpr int NVA = (NACT == NF) ? NVF : 0;
        v16i acc[L][NACT - NVA > 0 ? NACT - NVA : 1];
        v16i accv[L][NVA > 0 ? NVA : 1];
        v4i afx[KS - D > 0 ? KS - D : 1][L];                 // ring slots D .. KS-1 of a short visit (never live between visits)
        auto ring = [&](int slot, int l) -> v4i & { return slot < D ? afr[slot][l] : afx[slot - D][l]; };

        // ---- one 256-k stage.  FIRST: the accumulators start from the instruction's zero operand.
        auto stage = [&](auto first_tag, int kt) {
            constexpr bool FIRST = decltype(first_tag)::value;
            FQL_W4STAMP(ev++, 0);
            const char *sb = lds + (fs & 1) * C::W_STAGE;
            char *nb = lds + ((fs + 1) & 1) * C::W_STAGE;
            // this wave's fragment reads of the stage: one per-lane base per buffer, everything else immediate offsets
            // (formed per stage from an opaque copy: derived from the kernel-entry value they are loop invariants, hoisted
            //  and spilled)
            int rFo = rF0;
            asm volatile("" : "+v"(rFo));
            const char *sbf = sb + fbase * C::KFRAG + rFo, *nbf = nb + fbase * C::KFRAG + rFo;
            // what the loads of this stage fetch: weights two stages ahead, activations one stage ahead -- of this tile,
            // or of the next visit once this tile's K range is used up
            const bool w_here = kt + 2 < KT;
            const __amdgpu_buffer_rsrc_t rsW2 = w_here ? rsWc : rsWn;
            const int sW2b = w_here ? sWc : sWn;
            const int sW2 = sW2b == OOB ? OOB : sW2b + (w_here ? kt + 2 : kt + 2 - KT) * (FQL_KB / 2);
            const int nfr2 = w_here ? cur.nfr : nxt.nfr;
            const bool a_here = kt + 1 < KT;
            const int sA1b = a_here ? sAc : sAn;
            const int sA1 = sA1b == OOB ? OOB : sA1b + (a_here ? (kt + 1) * a_stage : 0);
            // scalar offsets of the activation refills, resolved once per stage (out of bounds stays out of bounds when a
            // step offset is added): a single wave has ~6 issue slots per matrix instruction, and a select + add per
            // load and step spent them
            int sAq[3][L];
            constexpr int RD = (NACT >= NF - 1 || W4_SHORT_RING == 0 || (W4_SHORT_RING == 2 && NACT != 2) || (W4_SHORT_RING == 3 && NACT != 3)) ? D : KS;        // ring depth of this tile class
            const bool last_stage = kt == KT - 1;
#pragma unroll
            for (int l = 0; l < L; ++l) {
                sAq[0][l] = sA0 == OOB ? OOB : sA0 + l * a_limb;
                sAq[1][l] = sA1 == OOB ? OOB : sA1 + l * a_limb;
                // the next stage's k-steps D .. : not fetched ahead of a visit boundary (ramp down to the contract)
                sAq[2][l] = (RD > D && last_stage) ? OOB : sAq[1][l];
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                // ---- the step as 18 SLOTS: one matrix instruction + at most one memory instruction + a few scalar / vector
                //      ones each, fenced so that the scheduler cannot move them.  Measured (tools/micro/w4_issue_probe.hip,
                //      ablation builds W4_ABLATE): at one wave per SIMD the next matrix instruction issues 32 cycles after
                //      the previous one only if what the wave issues in between takes <= ~24 cycles -- a 16-byte
                //      ds_read / buffer_load / ds_write ~16 each, a VALU / SALU / s_waitcnt ~4; three buffer loads in one
                //      gap cost 36 cycles of matrix pipe, a weight piece parked in one gap 75.
                //        fragment 0: this step's fragments 5 and 4 (their registers were busy until the end of the previous
                //                    step), activation refill limb 0           [at step 7: barrier after this fragment]
                //        fragment 1: next step's fragment 0, refills limb 1, 2
                //        fragment 2: next step's fragment 1, unpack dwords 0, 1 of weight piece ks
                //        fragment 3: next step's fragment 2, unpack dwords 2, 3
                //        fragment 4: next step's fragment 3, park the piece (2 ds_write_b128)
                //        fragment 5 (VGPR accumulators, skipped on narrow tiles): nothing; after it the piece's next load
                // (the swizzled address is recomputed per step from an opaque copy: hoisted, its 8 values cost 8 registers
                //  of a full register file and end up in scratch)
                const char *fc = sbf + ks * C::KSTRIDE;                                     // this step's fragments (from the wave's first)
                const char *fp = (ks == KS - 1) ? nbf : sbf + (ks + 1) * C::KSTRIDE;        // the coming step's
                const bool park = ks < NF && !(W4_ABLATE & 1);
                uint32_t up[8];
                auto mm = [&](int j, int l) {                // one matrix instruction of fragment j < NF - NVF
                    if (FIRST && ks == 0) {
                        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], ring(ks % RD, l), z, 0, 0, 0);
                    } else {
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], ring(ks % RD, l), acc[l][j], 0, 0, 0);
                    }
                };
                auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
                auto refill = [&](int l) {
                    // the ring slot the PREVIOUS step consumed gets the k-step D - 1 ahead of this one (of this stage, or of
                    // the one after it)
                    if (W4_ABLATE & 2) return;
                    const int pk = (ks + KS - 1) % KS, tk = ks - 1 + RD, w = tk < KS ? 0 : ((tk % KS) >= D ? 2 : 1);
                    ring(pk % RD, l) = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, sAq[w][l] + (tk % KS) * 1024, 0);
                };
                // first stage of a short visit: the ring holds k-steps 0 .. D-2 (the contract between visits: a D-deep visit
                // fetches k-step D-1 of its first stage in its first step, and so leaves slot D-1 to its successor); fetch
                // this stage's k-steps D-1 .. KS-2 on top of the regular refills (k-step KS-1 is step 0's regular refill)
                auto ramp_up = [&]() {
                    if constexpr (FIRST && RD > D) {
                        if (W4_ABLATE & 2) return;
#pragma unroll
                        for (int tk = (ks == 0 ? D - 1 : D + 2 * ks); tk < D + 2 * ks + 2 && tk < KS - 1; ++tk)
#pragma unroll
                            for (int l = 0; l < L; ++l)
                                ring(tk, l) = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, sAq[0][l] + tk * 1024, 0);
                    }
                };
                auto frag = [&](const char *base, int j) { if (!(W4_ABLATE & 4)) wf[j] = *reinterpret_cast<const v4i *>(base + j * C::KFRAG); };
                auto unpack = [&](int i) {
                    if (!park) return;
                    unpack8((uint32_t)bst[ks % NF][i], up[2 * i], up[2 * i + 1]);
                    asm volatile("" : "+v"(up[2 * i]), "+v"(up[2 * i + 1]));              // pin the unpack to this slot
                };
                static_assert(NF == 6 && L == 3 && NVF == 1, "the slot plans below are written for 6 fragments x 3 limbs, the last fragment in VGPRs");
                auto park0 = [&]() { if (park) *reinterpret_cast<v4i *>(nb + wA0 + ks * C::KFRAG) = v4i{(int)up[0], (int)up[1], (int)up[2], (int)up[3]}; };
                auto park1 = [&]() { if (park) *reinterpret_cast<v4i *>(nb + wA1 + ks * C::KFRAG) = v4i{(int)up[4], (int)up[5], (int)up[6], (int)up[7]}; };
                auto stage_barrier = [&]() {
                    if (ks == KS - 1) {
                        // every wave has parked the next stage (steps 0..NF-1) and holds the last fragments of this one
                        wait_lgkmcnt0();
                        __builtin_amdgcn_s_barrier();
                    }
                };
                if constexpr (NACT == 6) {
                    mm(0, 0); fence(); frag(fc, 5); fence();
                    mm(0, 1); fence(); frag(fc, 4); fence();
                    mm(0, 2); fence(); refill(0); fence();
                    stage_barrier();
                    mm(1, 0); fence(); frag(fp, 0); fence();
                    mm(1, 1); fence(); refill(1); fence();
                    mm(1, 2); fence(); refill(2); fence();
                    mm(2, 0); fence(); frag(fp, 1); fence();
                    mm(2, 1); fence(); unpack(0); fence();
                    mm(2, 2); fence(); unpack(1); fence();
                    mm(3, 0); fence(); frag(fp, 2); fence();
                    mm(3, 1); fence(); unpack(2); fence();
                    mm(3, 2); fence(); unpack(3); fence();
                    mm(4, 0); fence(); frag(fp, 3); fence();
                    mm(4, 1); fence(); park0(); fence();
                    mm(4, 2); fence(); park1(); fence();
                    // (tiles narrower than NF fragments take the 5-fragment body: no branch in this one)
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        if (FIRST && ks == 0) mfma_i8_vgpr_zero(accv[l][0], wf[NF - 1], afr[ks % D][l]);
                        else mfma_i8_vgpr(accv[l][0], wf[NF - 1], afr[ks % D][l]);
                    }
                } else if constexpr (NACT == 5) {
                    // 128-row tiles of at most 5 fragments (the column tiling deals N / 32 fragments over the tiles as evenly as
                    // possible: 5 and 6 at the headline shape): 15 matrix instructions, all accumulators in AGPRs, and
                    // exactly one memory / unpack slot behind each -- this step's fragment 4, the three refills, the coming
                    // step's fragments 0..3, the piece's four unpacks, its two parking stores, its next load
                    mm(0, 0); fence(); frag(fc, 4); fence();
                    mm(0, 1); fence(); refill(0); fence();
                    mm(0, 2); fence(); refill(1); fence();
                    stage_barrier();
                    mm(1, 0); fence(); frag(fp, 0); fence();
                    mm(1, 1); fence(); refill(2); fence();
                    mm(1, 2); fence(); unpack(0); fence();
                    mm(2, 0); fence(); frag(fp, 1); fence();
                    mm(2, 1); fence(); unpack(1); fence();
                    mm(2, 2); fence(); unpack(2); fence();
                    mm(3, 0); fence(); frag(fp, 2); fence();
                    mm(3, 1); fence(); unpack(3); fence();
                    mm(3, 2); fence(); park0(); fence();
                    mm(4, 0); fence(); frag(fp, 3); fence();
                    mm(4, 1); fence(); park1(); fence();
                    mm(4, 2); fence();
                } else if constexpr (NACT == 3) {            // 33..64 rows: 9 matrix instructions, the same staging work
                    mm(0, 0); fence(); frag(fc, 2); fence();
                    mm(0, 1); fence(); refill(0); fence();
                    mm(0, 2); fence(); refill(1); fence();
                    stage_barrier();
                    mm(1, 0); fence(); frag(fp, 0); fence();
                    mm(1, 1); fence(); refill(2); fence();
                    mm(1, 2); fence(); unpack(0); unpack(1); fence();
                    mm(2, 0); fence(); frag(fp, 1); fence();
                    mm(2, 1); fence(); unpack(2); unpack(3); fence();
                    mm(2, 2); fence(); park0(); fence();
                    park1();
                    ramp_up();
                } else {                                     // <= 32 rows: 6 matrix instructions
                    static_assert(NACT == 2, "tile classes: 6, 5, 3 or 2 fragments per wave");
                    mm(0, 0); fence(); frag(fc, 1); fence();
                    mm(0, 1); fence(); refill(0); fence();
                    mm(0, 2); fence(); refill(1); fence();
                    stage_barrier();
                    mm(1, 0); fence(); frag(fp, 0); fence();
                    mm(1, 1); fence(); refill(2); unpack(0); fence();
                    mm(1, 2); fence(); unpack(1); unpack(2); fence();
                    unpack(3); park0(); park1();
                    ramp_up();
                }
                if (park) bst[ks % NF] = __builtin_amdgcn_raw_buffer_load_b128(rsW2, voffW, ks < nfr2 ? sW2 + ks * pieceW : OOB, 0);   // (OOB + i * pieceW stays out of bounds)
                if (ks == NF && kt == KT - 1) park_sz(szbuf + (parity ^ 1) * 3 * C::BN);
                __builtin_amdgcn_sched_barrier(0);
            }
            ++fs;
            sA0 = sA1;
        };
        // ONE instance of the K loop: variants (per visit or per stage) that merge make the compiler copy all 288
        // accumulator registers out of the AGPRs at the merge.  A tile narrower than NF fragments skips the last
        // fragment's instructions through a wave-uniform branch in every k-step.
        stage(T_{}, 0);
        for (int kt = 1; kt < KT; ++kt) stage(F_{}, kt);
        FQL_W4STAMP(ev++, 0);

        // ---- epilogue: identical arithmetic to gemm_i8_kernel (the weights are the matrix instruction's A operand, so a
        //      lane owns ONE output row t and registers 4q..4q+3 are 4 consecutive output columns)
        // (per-lane values of the epilogue are derived from an opaque copy of the thread id: derived from the kernel-entry
        //  copies they are invariants of the persistent loop, hoisted out of it, live across the K loop and spilled)
        int tid_e = threadIdx.x;
        asm volatile("" : "+v"(tid_e));
        const int lane_e = tid_e & 63, l31_e = lane_e & 31, g_e = lane_e >> 5;
        const int rl_e = wm * FQL_MB + l31_e;
        const bool row_ok_e = active && rl_e < cur.rows_valid;
        const int t_e = row_ok_e ? cur.row0 + rl_e : 0;
        const float *sz = szbuf + parity * 3 * C::BN;
        // the matrix instructions' results have no interlock with the readers below (inline assembly on both sides)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        // One accumulator value as an integer in a VGPR.  The AGPR-resident accumulators are read by an explicit (volatile)
        // v_accvgpr_read_b32 where the value is used: left to itself the compiler copies all 240 AGPRs into VGPRs at the top
        // of the epilogue and spills the prefetch rings to make room.
        auto acc_val = [&](int l, int j, int r) -> int {
            if (NVA > 0 && j >= NF - NVF) return accv[l][0][r];
            int v;
            asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(acc[l][j < NF - NVF ? j : 0][r]));
            return v;
        };
        // mode 0: plain tile; 1: residual pass -- park the float32 results in this lane's scratch slot (workgroup-private,
        // read back by the same lane in the next visit); 2: main pass after a residual pass -- add the parked values.
        // The accumulators are read in ONE place per path (per-mode copies of a loop get their common accumulator reads
        // hoisted in front of the branch: all 288 at once).
        const int mode = (!RES || (!rpass && cur.ad == 0)) ? 0 : (rpass ? 1 : 2);
        const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & (out_kind == 0 ? 15 : 7)) == 0);
        float rs[L];
#pragma unroll
        for (int l = 0; l < L; ++l) rs[l] = (float)rsi[l];
        const int nfr_c = cur.nfr - fbase < 0 ? 0 : (cur.nfr - fbase < NACT ? cur.nfr - fbase : NACT);   // this wave's fragments that exist
        // nz4: the NEGATED zero points (park_sz).  tot = ((t_2 * 256 + t_1) * 256 + t_0, t_l = acc_l - zp * rowsum_l: the other
        // kernels write the first step as fmaf(0, 256, t_2) = t_2 + 0, which differs from t_2 only for t_2 = -0.0 -- and an
        // fma of a +0.0 / integer-valued addend with an exact-zero result is +0.0 under round-to-nearest: same bits, one add less.
        auto out4 = [&](int j, int q, const v4f &s4, const v4f &nz4, float (&o)[4]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float tot = fmaf(nz4[c], rs[L - 1], (float)acc_val(L - 1, j, 4 * q + c));
#pragma unroll
                for (int l = L - 2; l >= 0; --l)
                    tot = fmaf(tot, 256.0f, fmaf(nz4[c], rs[l], (float)acc_val(l, j, 4 * q + c)));
                o[c] = (tot * d) * s4[c];
            }
        };
        // (bias and row weight are applied on the way: one LDS read + add, one multiply per quad, wave-uniform branches)
        const bool fast = mode == 0 && vec && out_kind == 0 && cur.n0 + (fbase + nfr_c) * 32 <= N &&
                          (size_t)T * (size_t)N < ((size_t)1 << 29);      // (32-bit byte offsets into `out`)
        // the same for 16-bit outputs (float16 / bfloat16 rows out: one 8-byte store per quad)
        const bool fast16 = mode == 0 && vec && out_kind != 0 && cur.n0 + (fbase + nfr_c) * 32 <= N &&
                            (size_t)T * (size_t)N < ((size_t)1 << 29);
        if (row_ok_e && !(W4_ABLATE & 16)) {
            // One fragment (32 columns = 16 outputs of this lane) at a time: all its arithmetic as straight-line code, then
            // ONE branch on how to store -- the common case (float32 outputs, whole 16-byte stores, every column inside N)
            // or the general one.  (Two copies of the arithmetic, one per case, get their accumulator reads hoisted in
            // front of the branch by the compiler -- all 288 registers at once.)
            const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)((size_t)T * N * (out_kind == 0 ? 4 : 2)), 0x00020000);
            const int ovoff = (t_e * N + cur.n0 + fbase * 32 + 4 * g_e) * (out_kind == 0 ? 4 : 2);
            float *slot0 = res_scratch + ((size_t)blockIdx.x * C::NW + wave) * (NF * 1024) + lane_e * 4;
#pragma unroll
            for (int j = 0; j < NACT; ++j) {
                __builtin_amdgcn_sched_barrier(0);           // (register pressure: no fragment's reads before its turn)
                if (j >= nfr_c) continue;
                if (ev <= 21) FQL_W4STAMP(42 + j, 0);        // (trace builds: the first visit's epilogue, fragment by fragment)
                float o[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = (fbase + j) * 32 + 8 * q + 4 * g_e;
                    const v4f s4 = *reinterpret_cast<const v4f *>(sz + c0);
                    const v4f z4 = *reinterpret_cast<const v4f *>(sz + C::BN + c0);
                    out4(j, q, s4, z4, o[q]);
                }
                if (fast || fast16) {                        // (the same order as the general path below: + bias, then x row weight)
                    if (bias != nullptr) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const v4f b4 = *reinterpret_cast<const v4f *>(sz + 2 * C::BN + (fbase + j) * 32 + 8 * q + 4 * g_e);
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[q][c] += b4[c];
                        }
                    }
                    if (row_scaled) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[q][c] *= rw;
                    }
                }
                if (fast) {
                    // (buffer stores: ONE 32-bit offset register per lane, fragment and quad in the scalar offset -- 64-bit
                    //  store pointers kept across the fragments were spilled, and every scratch reload is a vmcnt(0) wait
                    //  behind the next tile's prefetch)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_raw_buffer_store_b128(v4i{__builtin_bit_cast(int, o[q][0]), __builtin_bit_cast(int, o[q][1]),
                                                                   __builtin_bit_cast(int, o[q][2]), __builtin_bit_cast(int, o[q][3])},
                                                               rsO, ovoff, (j * 32 + 8 * q) * 4, 0);
                } else if (fast16) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        unsigned short h[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) h[c] = (out_kind == 1) ? f32_to_f16_bits(o[q][c]) : f32_to_bf16_bits(o[q][c]);
                        __builtin_amdgcn_raw_buffer_store_b64(v2i{(int)((uint32_t)h[0] | ((uint32_t)h[1] << 16)), (int)((uint32_t)h[2] | ((uint32_t)h[3] << 16))},
                                                              rsO, ovoff, (j * 32 + 8 * q) * 2, 0);
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c0 = (fbase + j) * 32 + 8 * q + 4 * g_e;
                        if (RES && mode == 1) {
                            *reinterpret_cast<v4f *>(slot0 + (j * 4 + q) * 256) = v4f{o[q][0], o[q][1], o[q][2], o[q][3]};
                        } else {
                            if (RES && mode == 2 && addp) {  // rows without a residual stay bit-identical to mode 0
                                const v4f pr = *reinterpret_cast<const v4f *>(slot0 + (j * 4 + q) * 256);
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[q][c] += pr[c];
                            }
                            if (bias != nullptr) {
                                const v4f b4 = *reinterpret_cast<const v4f *>(sz + 2 * C::BN + c0);
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[q][c] += b4[c];
                            }
                            if (row_scaled) {
#pragma unroll
                                for (int c = 0; c < 4; ++c) o[q][c] *= rw;
                            }
                            store_out4(out, out_kind, (size_t)t_e * N, cur.n0 + c0, N, vec, o[q]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // (after the epilogue, not under it: 16 registers held across the epilogue's 200+ live ones were spilled)
        read_first_frags(lds + (fs & 1) * C::W_STAGE, wave_fbase(tile_class(nxt)));
        };   // visit_body
        // (one copy of K loop + epilogue per tile class, each with its own accumulators: a fragment skipped under a branch
        //  inside ONE copy turns the accumulators into phi nodes and the compiler then spills whole tuples around the epilogue)
        if (cls == 4 && cur.nfr == NF) visit_body(std::integral_constant<int, 6>{});
        else if (cls == 4) visit_body(std::integral_constant<int, 5>{});
        else if (cls == 2) visit_body(std::integral_constant<int, 3>{});
        else visit_body(std::integral_constant<int, 2>{});
        FQL_W4STAMP(ev++, 0);
        FQL_W4STAMP(ev++, 1);
        if (!nxt.nt) break;                                  // past the last tile of this workgroup
        cur = nxt;
        parity ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}
