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

#ifndef FQL_R32_W_AUX
#define FQL_R32_W_AUX 2        // cache policy of the weight-stream loads: nt (read once; -2 % measured)
#endif

template <int L, int NF, int KG, int DEPTH, int BDEPTH, int OCC = 2>
struct Rows32Cfg {
    static constexpr int NW = 8;
    static constexpr int NG = NW / KG;                        // column groups of the workgroup
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = FQL_MB;
    static constexpr int BN = 32 * NF * NG;
    static constexpr int KS = FQL_KB / 32;
    static constexpr int D = DEPTH;                           // A prefetch depth in k-steps
    static constexpr int BD = BDEPTH;                         // weight stages in flight per wave (register ring)
    static constexpr int WG_PER_CU = OCC / 2;                 // OCC waves per SIMD = OCC / 2 workgroups of 8 waves per CU
    static constexpr int PIECES = NF * 4;                     // 1 KiB weight pieces per wave per 256-k stage
    static constexpr int SLAB = NF * 32 * (FQL_KB / 2);       // wave-private LDS bytes (one stage of packed weights)
    static constexpr int ACC_BYTES = L * NF * 16 * 64 * 4;    // one wave's accumulators
    static constexpr int RED_BYTES = (KG > 1) ? (KG / 2) * NG * ACC_BYTES : 0;   // first round of the K-group tree
    static constexpr int MAIN_BYTES = (NW * SLAB > RED_BYTES) ? NW * SLAB : RED_BYTES;
    static constexpr int SZ_BYTES = 2 * 3 * BN * 4;           // scale / zero-point / bias slices of two tiles
    static constexpr int LDS_BYTES = MAIN_BYTES + SZ_BYTES;
    static constexpr int SZN = (3 * BN + THREADS - 1) / THREADS;
    static_assert(KG == 1 || KG == 2 || KG == 4 || KG == 8, "K split");
    static_assert(KS % D == 0, "ring depth must divide the steps per stage");
    static_assert(LDS_BYTES * WG_PER_CU <= 160 * 1024, "LDS budget");
};

#if defined(FQL_TRACE)
__device__ unsigned long long fql_trace_buf[8 * 64];         // [block < 8][event]: s_memtime stamps of wave 0
#define FQL_STAMP(i) do { if (blockIdx.x < 8 && threadIdx.x == 0 && (i) < 64) fql_trace_buf[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FQL_STAMP(i) do { } while (0)
#endif
#if defined(FQL_TRACE) && FQL_TRACE >= 2                      // finer stamps inside a stage (rows16)
#define FQL_STAMP_FINE(i) FQL_STAMP(i)
#else
#define FQL_STAMP_FINE(i) do { } while (0)
#endif

typedef GemmTile Rows32Tile;   // wave-uniform description of one visit of a 32-row x BN tile (fql_gemm_i8.h)

template <int L, int NF, int KG, int DEPTH, int BDEPTH, int OCC>
__global__ __launch_bounds__(512, OCC) void gemm_i8_rows32_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, void *__restrict__ out, int out_kind_flags,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles, int m_slots, float *__restrict__ res_scratch,
    const float *__restrict__ bias)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Rows32Cfg<L, NF, KG, DEPTH, BDEPTH, OCC>;
    // out_kind_flags: bits 0-1 the output element type (FQL_DTYPE_*), bit 3: multiply every output row by its row weight
    // (plane delta[sets * T + t], written by the pre-pass of fql_moe_gather_scaled_fwd_f32: the routing weight folded into
    // the epilogue, so that the combine step is a pure gather-add).  One rounding, after the bias: (x W^T + b) * w.
    const int out_kind = out_kind_flags & 3;
    const bool row_scaled = (out_kind_flags & 8) != 0;
    constexpr bool RES = FQL_RES_ENABLED && (L >= 2);                           // residual limb set for heavy-tailed rows (fql_gemm_i8.h)
    constexpr int KS = C::KS, D = C::D, NG = C::NG, BD = C::BD;
    constexpr int OOB = 0x7fff0000;                           // a buffer offset past every descriptor: reads zero
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NG, ng = wave - kg * NG;            // this wave's K group and column group
    const int l31 = lane & 31, g = lane >> 5;

    // ---- expert table: ONE pass over the device-side counts per workgroup, kept in LDS (first 64 experts; more
    //      experts fall back to re-reading per tile): finding a tile's expert later costs no global load
    __shared__ int s_lo[64], s_cnt[64], s_tex[64], s_pex[64];
    int n_real = m_slots * n_tiles;
    if (tpe != nullptr) {
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) {
            const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
            if (base == 0 && wave == 0) { s_lo[lane] = x.lo; s_cnt[lane] = x.cnt; s_tex[lane] = x.tile_excl; s_pex[lane] = x.pad_excl; }
        }
        const int m_tiles = ct < m_slots ? ct : m_slots;
        n_real = m_tiles * n_tiles;
    }
    n_real = __builtin_amdgcn_readfirstlane(n_real);
    __syncthreads();

    // ---- tile id -> (expert, 32-row block, column block); m-tile major so neighbours share activations in L2
    auto tile_params = [&](int vb) -> Rows32Tile {
        Rows32Tile tp = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vb >= n_real) return tp;
        const int tile = xcd_remap(vb, n_real);
        const int ms = tile / n_tiles;
        tp.nt = tile - ms * n_tiles;
        if (tpe == nullptr) {
            tp.row0 = tp.prow0 = ms * C::BM;
            tp.rows_valid = T - tp.row0;
            tp.ok = 1;
        } else {
            {   // first 64 experts from LDS (lane i looks at expert i)
                const int lo = s_lo[lane], cnt = s_cnt[lane], te = s_tex[lane], pe = s_pex[lane];
                const int tiles_e = (cnt + C::BM - 1) / C::BM;
                const unsigned long long hit = __ballot(lane < E && ms >= te && ms < te + tiles_e);
                if (hit) {
                    const int src = __ffsll((long long)hit) - 1;
                    const int lo_s = wave_bcast(lo, src), cnt_s = wave_bcast(cnt, src);
                    const int te_s = wave_bcast(te, src), pe_s = wave_bcast(pe, src);
                    tp.e = src;
                    tp.row0 = lo_s + (ms - te_s) * C::BM;
                    tp.prow0 = pe_s + (ms - te_s) * C::BM;
                    tp.rows_valid = cnt_s - (ms - te_s) * C::BM;
                    tp.ok = 1;
                }
            }
            if (!tp.ok && E > 64) {
                int cp = 0, ct = 0;
                for (int base = 0; base < E && !tp.ok; base += 64) {
                    const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
                    const unsigned long long hit = __ballot(ms >= x.tile_excl && ms < x.tile_excl + x.tiles);
                    if (hit && base > 0) {
                        const int src = __ffsll((long long)hit) - 1;
                        const int lo = wave_bcast(x.lo, src), cnt = wave_bcast(x.cnt, src);
                        const int te = wave_bcast(x.tile_excl, src), pe = wave_bcast(x.pad_excl, src);
                        tp.e = base + src;
                        tp.row0 = lo + (ms - te) * C::BM;
                        tp.prow0 = pe + (ms - te) * C::BM;
                        tp.rows_valid = cnt - (ms - te) * C::BM;
                        tp.ok = 1;
                    }
                }
            }
        }
        if (tp.rows_valid <= 0) tp.ok = 0;
        if (tp.rows_valid > C::BM) tp.rows_valid = C::BM;
        tp.e = __builtin_amdgcn_readfirstlane(tp.e);
        tp.row0 = __builtin_amdgcn_readfirstlane(tp.row0);
        tp.prow0 = __builtin_amdgcn_readfirstlane(tp.prow0);
        tp.rows_valid = __builtin_amdgcn_readfirstlane(tp.rows_valid);
        tp.nt = __builtin_amdgcn_readfirstlane(tp.nt);
        tp.ok = __builtin_amdgcn_readfirstlane(tp.ok);
        return tp;
    };

    const int KB = Kp / FQL_KB, KT = KB;
    const int SP = ((KT + KG - 1) / KG + BD - 1) / BD * BD;   // stages per wave per tile, padded to the ring depth
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)(RES ? 2 : 1) * L * KB * MBT * 8192), 0x00020000);
    const int a_stage = MBT * 8192;
    const int a_limb = KB * MBT * 8192;                       // one limb plane; the residual set starts L planes in

    // ---- per-lane offsets that do not depend on the tile: everything tile-specific goes into the scalar offset
    int aoffl[L];
#pragma unroll
    for (int l = 0; l < L; ++l) aoffl[l] = (l * KB) * MBT * 8192 + lane * 16;
    char *slab = lds + wave * C::SLAB;
    int vrel[C::PIECES], wB[C::PIECES];                       // piece i = rows 8i..8i+7 of this wave's NF*32 rows
#pragma unroll
    for (int i = 0; i < C::PIECES; ++i) {
        const int row = i * 8 + (lane >> 3), ch = lane & 7;
        vrel[i] = row * (K >> 1) + ch * 16;
        wB[i] = row * 128 + 16 * (ch ^ ((row >> 1) & 7));
    }
    int rB[NF], swB[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const int n = j * 32 + l31;
        rB[j] = n * 128;
        swB[j] = (n >> 1) & 7;
    }
    // scalar offsets of (tile, stage index s of this K group [, k-step]); past the tile's K range or for a
    // missing tile the weights read as zero (the activations may then be anything)
    auto w_soff = [&](const Rows32Tile &tp, int s) -> int {
        const int kt = kg + s * KG;
        return (tp.ok && kt < KT) ? (tp.nt * C::BN + ng * NF * 32) * (K >> 1) + kt * (FQL_KB / 2) : OOB;
    };
    auto a_soff = [&](const Rows32Tile &tp, int s, int ks) -> int {
        const int kt = kg + s * KG;
        return (tp.ok && kt < KT) ? (tp.prow0 >> 5) * 8192 + kt * a_stage + ks * 1024 + (tp.rp ? L * a_limb : 0) : 0;
    };
    auto w_rsrc = [&](const Rows32Tile &tp) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(packed + (size_t)tp.e * wbytes), 0, (int)wbytes, 0x00020000);
    };

    // ---- rings: BD weight stages and D activation k-steps in flight, running ACROSS tile boundaries so that a
    //      tile's reduction and epilogue hide the next tile's first HBM round trip
    v4i bst[BD][C::PIECES];
    v4i afr[D][L];
    float szr[C::SZN];
    int drow[1 + L];                                          // delta bits and limb row sums of this lane's row
    int rwbits = 0;                                           // this lane's row weight (row_scaled)
    int d2bits = 0;                                           // delta2 bits of this lane's row (heavy-tailed rows: fql_gemm_i8.h)
    float *szbuf = reinterpret_cast<float *>(lds + C::MAIN_BYTES);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)delta, 0, ((RES ? 2 : 1) + 1) * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)rowsum, 0, (RES ? 2 : 1) * L * T * 4, 0x00020000);
    // the tile's scale / zero-point slice and this lane's row values travel with the tile's first loads, so the
    // epilogue issues no global load (a load there would queue behind the next tile's HBM prefetch)
    auto issue_tile_consts = [&](const Rows32Tile &tp) {
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(scales + (size_t)tp.e * N), 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(zps + (size_t)tp.e * N), 0, N * 4, 0x00020000);
const __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(bias != nullptr ? bias + (size_t)tp.e * N : scales), 0, bias != nullptr ? N * 4 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::SZN; ++i) {
            const int idx = tid + i * C::THREADS;             // < BN: scale; < 2 BN: zero point; then the bias
            const int arr = idx < C::BN ? 0 : (idx < 2 * C::BN ? 1 : 2);
            const int col = idx - arr * C::BN;
            const int so = (tp.ok && idx < 3 * C::BN) ? 0 : OOB;
            const int vo = (tp.nt * C::BN + col) * 4;
            const int vs = __builtin_amdgcn_raw_buffer_load_b32(rsS, arr == 0 ? vo : OOB, so, 0);
            const int vz = __builtin_amdgcn_raw_buffer_load_b32(rsZ, arr == 1 ? vo : OOB, so, 0);
            const int vb = __builtin_amdgcn_raw_buffer_load_b32(rsBi, arr == 2 ? vo : OOB, so, 0);
            szr[i] = __builtin_bit_cast(float, vs | vz | vb);
        }
        const int t = (tp.ok && l31 < tp.rows_valid) ? tp.row0 + l31 : 0;
        const int tsel = tp.rp ? T : 0;                       // residual pass: the second set of per-row values
        drow[0] = __builtin_amdgcn_raw_buffer_load_b32(rsD, (tsel + t) * 4, 0, 0);
#pragma unroll
        for (int l = 0; l < L; ++l) drow[1 + l] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (L * tsel + l * T + t) * 4, 0, 0);
        if (RES) d2bits = __builtin_amdgcn_raw_buffer_load_b32(rsD, (T + t) * 4, 0, 0);
        if (row_scaled) rwbits = __builtin_amdgcn_raw_buffer_load_b32(rsD, ((RES ? 2 : 1) * T + t) * 4, 0, 0);
    };
    int ev = 0; (void)ev;
    FQL_STAMP(ev++);                                          // 0: kernel entry (after n_real)
    Rows32Tile cur = tile_params(blockIdx.x);
    if constexpr (RES) cur.rp = tile_has_residual(delta, T, cur, C::BM, lane) && res_scratch != nullptr;
    FQL_STAMP(ev++);                                          // 1: first tile params
    {
        const __amdgpu_buffer_rsrc_t rs = w_rsrc(cur);
#pragma unroll
        for (int u = 0; u < BD; ++u) {
            const int so = w_soff(cur, u);
#pragma unroll
            for (int i = 0; i < C::PIECES; ++i) bst[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vrel[i], so, FQL_R32_W_AUX);
        }
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int so = a_soff(cur, 0, d);
#pragma unroll
            for (int l = 0; l < L; ++l) afr[d][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoffl[l], so, 0);
        }
        issue_tile_consts(cur);
    }
    int parity = 0;

  for (int vb = blockIdx.x; vb < n_real; parity ^= 1) {     // one iteration per visit (tile, pass): see GemmTile
    FQL_STAMP(ev++);                                          // tile: start
    Rows32Tile nxt;
    const bool rpass = RES && cur.rp != 0;
    if (rpass) { nxt = cur; nxt.rp = 0; nxt.ad = 1; }
    else { vb += gridDim.x; nxt = tile_params(vb); }
    // heavy-tail probe of the next tile: issued now, evaluated at the start of the last stage (where nxt is first used)
    ResidualProbe pb = {0, 0};
    if constexpr (RES) pb = residual_probe_issue(delta, T, nxt, C::BM, lane, res_scratch != nullptr && !rpass);
    FQL_STAMP(ev++);                                          // tile: next params known
    const __amdgpu_buffer_rsrc_t rs_cur = w_rsrc(cur), rs_nxt = w_rsrc(nxt);
    float *sz = szbuf + parity * 3 * C::BN;
#pragma unroll
    for (int i = 0; i < C::SZN; ++i)
        if (tid + i * C::THREADS < 3 * C::BN) sz[tid + i * C::THREADS] = szr[i];
    const float d = __builtin_bit_cast(float, drow[0]);
    const float rw = __builtin_bit_cast(float, rwbits);
    const bool addp = RES && (d2bits & 0x7fffffff) != 0;
    float rsum[L];
#pragma unroll
    for (int l = 0; l < L; ++l) rsum[l] = (float)drow[1 + l];

    v16i acc[L][NF];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;

    // ---- K loop over this K group's stages kg, kg+KG, ...: no barriers, wave-local ordering only
    for (int s0 = 0; s0 < SP; s0 += BD) {
#pragma unroll
      for (int u = 0; u < BD; ++u) {                          // unrolled: the ring slot is static
        const int s = s0 + u;
        FQL_STAMP(ev++);                                      // tile: stage start
        // LDS operations of one wave execute in order: these writes cannot overtake the previous stage's
        // fragment reads
#pragma unroll
        for (int i = 0; i < C::PIECES; ++i) *reinterpret_cast<v4i *>(slab + wB[i]) = bst[u][i];
        if constexpr (RES) { if (s + 1 == SP && !rpass) nxt.rp = residual_probe_eval(pb); }
        if (s + 1 == SP) issue_tile_consts(nxt);              // before the younger HBM loads of this boundary
        {   // refill the slot with the stage BD ahead (the next tile's first stages near the end of this one)
            const bool here = s + BD < SP;
            const int so = here ? w_soff(cur, s + BD) : w_soff(nxt, s + BD - SP);
            if (here) {
#pragma unroll
                for (int i = 0; i < C::PIECES; ++i) bst[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_cur, vrel[i], so, FQL_R32_W_AUX);
            } else {
#pragma unroll
                for (int i = 0; i < C::PIECES; ++i) bst[u][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_nxt, vrel[i], so, FQL_R32_W_AUX);
            }
        }
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
#if defined(FQL_ABLATE) && FQL_ABLATE == 1          // timing experiments only (wrong results)
#pragma unroll
            for (int l = 0; l < L; ++l) asm volatile("" ::"v"(afr[ks % D][l]));
#pragma unroll
            for (int j = 0; j < NF; ++j) asm volatile("" ::"v"(bfr[j]));
#else
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr[j], afr[ks % D][l], acc[l][j], 0, 0, 0);
#endif
            {   // refill the A ring slot D steps ahead (next stage / next tile near the end)
                const int nks = ks + D;
                int so;
                if (nks < KS) so = a_soff(cur, s, nks);
                else so = (s + 1 < SP) ? a_soff(cur, s + 1, nks - KS) : a_soff(nxt, 0, nks - KS);
#pragma unroll
                for (int l = 0; l < L; ++l) afr[ks % D][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoffl[l], so, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

    FQL_STAMP(ev++);                                          // tile: K loop done
    // ---- add the KG partial accumulators through LDS, pairwise (int32: exact, order-free).  The weight slabs
    //      are dead by the first barrier; the next tile's first loads are already in flight.
    if (KG > 1) {
#pragma unroll
        for (int sft = 1; sft < KG; sft <<= 1) {
            char *red = lds + ((kg / (2 * sft)) * NG + ng) * C::ACC_BYTES;
            __syncthreads();                                  // slabs / previous round's partials consumed
            if ((kg & (2 * sft - 1)) == sft) {
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
            if ((kg & (2 * sft - 1)) == 0) {
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
    } else {
        __syncthreads();                                      // the scale slice parked by other waves
    }

    FQL_STAMP(ev++);                                          // tile: reduction done
    // ---- epilogue (as the wide kernel): lane owns output row t, registers 4q..4q+3 are 4 consecutive columns
    const Rows32Tile done = cur;
    cur = nxt;
    if (kg != 0 || !done.ok || l31 >= done.rows_valid) continue;
    const int t = done.row0 + l31;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & (out_kind == 0 ? 15 : 7)) == 0);
    // MODE 0 plain tile / 1 residual pass (park float32 results in the scratch slot) / 2 main pass after it (add them)
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        float *slot0 = (MODE == 0) ? nullptr : res_scratch + ((size_t)blockIdx.x * 8 + wave) * (NF * 1024) + lane * 4;
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = (ng * NF + j) * 32 + 8 * q + 4 * g;            // column inside the tile
                const v4f s4 = *reinterpret_cast<const v4f *>(sz + c0);
                const v4f z4 = *reinterpret_cast<const v4f *>(sz + C::BN + c0);
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tot = 0.0f;
#pragma unroll
                    for (int l = L - 1; l >= 0; --l)
                        tot = fmaf(tot, 256.0f, fmaf(-z4[c], rsum[l], (float)acc[l][j][4 * q + c]));
                    o[c] = (tot * d) * s4[c];
                }
                if constexpr (MODE == 1) {
                    *reinterpret_cast<v4f *>(slot0 + (j * 4 + q) * 256) = v4f{o[0], o[1], o[2], o[3]};
                } else {
                    if constexpr (MODE == 2) {
                        if (addp) {
                            const v4f pr = *reinterpret_cast<const v4f *>(slot0 + (j * 4 + q) * 256);
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] += pr[c];
                        }
                    }
                    if (bias != nullptr) {
                    const v4f b4 = *reinterpret_cast<const v4f *>(sz + 2 * C::BN + c0);
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] += b4[c];
                }
                if (row_scaled) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] *= rw;
                }
                store_out4(out, out_kind, (size_t)t * N, done.nt * C::BN + c0, N, vec, o);
                }
            }
    };
    if (!RES || (done.rp == 0 && done.ad == 0)) epilogue(std::integral_constant<int, 0>{});
    else if (done.rp) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 2>{});
  }
#endif
}
