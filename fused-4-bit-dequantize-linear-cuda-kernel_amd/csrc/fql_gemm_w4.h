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
// LDS image of a weight stage (unpacked, 256 B per weight row = 8 k-steps x 2 lane groups x 16 B):
//   row n of the tile, slot s = 8 (ks & 1) + 2 (ks >> 1) + g  ->  byte n * 256 + 16 * (s ^ (n & 15))
// conflict-free for the ds_read_b128 fragment reads (64 banks; a 16-lane group holds 16 different n & 15, hence 16
// different slots) and for the parking ds_write_b128s (32 banks, groups of 8 lanes = the 8 chunks c of one row, which go
// to slots c and 8 + c: 8 different slots mod 8 per instruction; slot 2 ks + g, the first layout, put chunks c and
// c + 4 on the same banks: SQ_LDS_BANK_CONFLICT 8 cycles per store).
//
// Replaces (reference, CUDA): csrc/moe_int4_kernel.cu:17-136 and csrc/quantized_linear_kernel.cu:90-279.
#pragma once
#include "fql_common.h"
#include "fql_gemm_i8.h"

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

// byte offset (XOR-ed into a lane's fragment address) of k-step ks inside a weight row's 256-byte stage image
__device__ __host__ constexpr int frag_xor(int ks) { return 16 * (8 * (ks & 1) + 2 * (ks >> 1)); }

template <int L, int NF, int DEPTH = 8>
struct W4Cfg {
    static constexpr int D = DEPTH;                          // activation ring depth in k-steps (8 = a full stage)
    static constexpr int NW = 4;
    static constexpr int THREADS = 256;
    static constexpr int BM = 4 * FQL_MB;                    // 128 rows: one 32-row block per wave
    static constexpr int BN = 32 * NF;
    static constexpr int KS = FQL_KB / 32;                   // 8 k-steps per stage
    static constexpr int W_STAGE = BN * FQL_KB;              // bytes of UNPACKED weights per stage
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
#ifndef W4_ABLATE
#define W4_ABLATE 0
#endif
#if defined(FQL_TRACE)
__device__ unsigned long long fql_trace_w4[8 * 64];
#define FQL_W4STAMP(i, real) do { if (blockIdx.x < 8 && threadIdx.x == 0 && (i) < 64) fql_trace_w4[blockIdx.x * 64 + (i)] = (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FQL_W4STAMP(i, real) do { } while (0)
#endif

template <int L, int NF, int DEPTH>
__global__ __launch_bounds__(256, 1) void gemm_w4_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, void *__restrict__ out, int out_kind,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles_min, int m_slots, float *__restrict__ res_scratch,
    const float *__restrict__ bias, int n_tiles_alt, int part)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using C = W4Cfg<L, NF, DEPTH>;
    constexpr int KS = C::KS, D = C::D;
    // The compiler gives the matrix instruction's builtin AGPR accumulators (256 registers: 16 tiles of 16); the
    // fragments past that budget accumulate in VGPRs through the instruction's VGPR form (inline assembly).
    constexpr int NVF = (L * NF * 16 > 256) ? NF - 256 / (L * 16) : 0;
    constexpr bool RES = FQL_RES_ENABLED && (L >= 2);
    constexpr int OOB = 0x7fff0000;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    const int wm = wave;                                     // this wave's 32-row block of every tile

    // ---- column tiling: exactly the wide kernel's (fql_gemm_i8.h): the N / 32 fragments of a row block dealt as evenly
    //      as possible over the tile count picked here from the real row-block count
    int n_tiles = n_tiles_min;
    int n_real = m_slots * n_tiles;
    const int n_frag = (N + 31) >> 5;
    auto pick_tiles = [&](int m_tiles) {
        if (n_tiles_alt <= 0) return;
        const int G = (int)gridDim.x;
        const float ra = (float)((m_tiles * n_tiles_min + G - 1) / G), rb = (float)((m_tiles * n_tiles_alt + G - 1) / G);
        const float ca = ra * (float)(n_frag + n_tiles_min) * (float)n_tiles_alt;
        const float cb = rb * (float)(n_frag + n_tiles_alt) * (float)n_tiles_min;
        if (cb < ca) n_tiles = n_tiles_alt;
    };
    if (tpe != nullptr) {
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) (void)expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct, part);
        const int m_tiles = __builtin_amdgcn_readfirstlane(ct < m_slots ? ct : m_slots);
        pick_tiles(m_tiles);
        n_real = m_tiles * n_tiles;
    } else {
        pick_tiles(m_slots);
        n_real = m_slots * n_tiles;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    n_real = __builtin_amdgcn_readfirstlane(n_real);
    if ((int)blockIdx.x >= n_real) return;
    const int f_base = n_frag / n_tiles, f_rem = n_frag - f_base * n_tiles;

    auto tile_params = [&](int vb) -> GemmTile {
        GemmTile tp = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (vb >= n_real) return tp;
        const int tile = xcd_remap(vb, n_real);
        const int ms = tile / n_tiles;
        tp.nt = tile - ms * n_tiles;
        tp.nfr = f_base + (tp.nt < f_rem ? 1 : 0);
        tp.n0 = (tp.nt * f_base + (tp.nt < f_rem ? tp.nt : f_rem)) * 32;
        if (tpe == nullptr) {
            tp.row0 = tp.prow0 = ms * C::BM;
            tp.rows_valid = T - tp.row0;
            tp.ok = 1;
        } else {
            int cp = 0, ct = 0;
            for (int base = 0; base < E && !tp.ok; base += 64) {
                const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct, part);
                const unsigned long long hit = __ballot(ms >= x.tile_excl && ms < x.tile_excl + x.tiles);
                if (hit) {
                    const int src = __ffsll((long long)hit) - 1;
                    const int lo = __shfl(x.lo, src, 64), cnt = __shfl(x.cnt, src, 64);
                    const int te = __shfl(x.tile_excl, src, 64), pe = __shfl(x.pad_excl, src, 64);
                    tp.e = base + src;
                    tp.row0 = lo + (ms - te) * C::BM;
                    tp.prow0 = pe + (ms - te) * C::BM;
                    tp.rows_valid = cnt - (ms - te) * C::BM;
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
    auto fill_table = [&](int first) {
        for (int i = wave; i < C::NTAB; i += C::NW) {
            const long long vbl = (long long)blockIdx.x + (long long)(first + i) * (long long)gridDim.x;
            GemmTile tp = tile_params(vbl < (long long)n_real ? (int)vbl : n_real);
            int hr = 0;
            if constexpr (RES) {
                if (res_scratch != nullptr && tp.ok) hr = tile_has_residual(delta, T, tp, C::BM, lane);
            }
            if (lane == 0) {
                int *p = tab + i * C::TAB_INTS;
                p[0] = tp.e; p[1] = tp.row0; p[2] = tp.prow0; p[3] = tp.rows_valid;
                p[4] = tp.n0; p[5] = tp.nfr; p[6] = tp.ok | (vbl < (long long)n_real ? 2 : 0); p[7] = hr;
            }
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
    const int wA0 = rowW * 256 + 16 * (chW ^ (rowW & 15));
    const int wA1 = rowW * 256 + 16 * ((chW + 8) ^ (rowW & 15));
    // fragment reads: row n = 32 j + l31, slot 8 (ks & 1) + 2 (ks >> 1) + g  ->  (l31 * 256 + 16 * (g ^ (l31 & 15))) ^ frag_xor(ks),
    // + j * 8192
    const int rF0 = l31 * 256 + 16 * (g ^ (l31 & 15));
    float *szbuf = reinterpret_cast<float *>(lds + 2 * C::W_STAGE);

    v4i bst[NF];                                             // one packed weight stage in flight (global -> VGPR)
    v4i afr[D][L];                                           // activation ring: D k-steps ahead (8: one full stage)
    v4i wf[NF];                                              // weight fragments of the coming k-step
    float szr[C::SZN];

    auto weight_rsrc = [&](int e) -> __amdgpu_buffer_rsrc_t {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);
    };
    // scalar offsets of a tile's operands (OOB: the loads read zero and move nothing)
    auto w_base = [&](const GemmTile &tp) -> int { return tp.ok ? tp.n0 * (K >> 1) : OOB; };
    auto a_base = [&](const GemmTile &tp) -> int {
        const bool act = tp.ok && wm * FQL_MB < tp.rows_valid;
        return act ? ((tp.prow0 >> 5) + wm) * 8192 + ((RES && tp.rp) ? L * a_limb : 0) : OOB;
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
        *reinterpret_cast<v4i *>(buf + wA0 + i * 8192) = v4i{(int)lo0, (int)hi0, (int)lo1, (int)hi1};
        *reinterpret_cast<v4i *>(buf + wA1 + i * 8192) = v4i{(int)lo2, (int)hi2, (int)lo3, (int)hi3};
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
    auto park_sz = [&](float *sz) {
#pragma unroll
        for (int i = 0; i < C::SZN; ++i)
            if (tid + i * C::THREADS < 3 * C::BN) sz[tid + i * C::THREADS] = szr[i];
    };
    auto read_frags = [&](const char *buf, int ks) {         // all NF fragments of k-step ks
        const char *p = buf + (rF0 ^ frag_xor(ks));
#pragma unroll
        for (int j = 0; j < NF; ++j) wf[j] = *reinterpret_cast<const v4i *>(p + j * 8192);
    };

    // ---- kernel prologue: the state every visit starts from
    //        LDS buffer fs & 1 holds stage 0 of the visit, bst its stage 1 (in flight), afr the A fragments of its
    //        stage 0 (in flight), wf the weight fragments of its k-step 0, sz[parity] its scale / zero-point slice
    fill_table(0);
    int ti = 0;                                              // tile index of this workgroup (table slot ti % NTAB)
    GemmTile cur = load_tile(0);
    if constexpr (!RES) cur.rp = 0;
    int fs = 0;                                              // stages since kernel start: LDS buffer parity
    int parity = 0;
    {
        const __amdgpu_buffer_rsrc_t rs = weight_rsrc(cur.e);
        const int sW = w_base(cur);
        issue_weights(rs, sW, cur.nfr);
        issue_sz(cur);
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
        read_frags(lds, 0);
    }

    int ev = 0; (void)ev;
    for (;;) {                                               // one iteration per VISIT (tile, pass)
        FQL_W4STAMP(ev++, 1);
        FQL_W4STAMP(ev++, 0);
        const bool rpass = RES && cur.rp != 0;
        const bool active = cur.ok && wm * FQL_MB < cur.rows_valid;
        // ---- the visit after this one: the main pass of the same tile after its residual pass, else the next tile
        GemmTile nxt;
        if (rpass) { nxt = cur; nxt.rp = 0; nxt.ad = 1; }
        else {
            ++ti;
            if ((ti % C::NTAB) == 0) fill_table(ti);         // (rare: more than 16 tiles per workgroup; drains the pipeline)
            nxt = load_tile(ti % C::NTAB);
            if constexpr (!RES) nxt.rp = 0;
        }
        // ---- per-row values of this visit's epilogue and the next visit's scale / zero-point slice: issued now, used
        //      a whole K loop later
        const int rl = wm * FQL_MB + l31;
        const bool row_ok = active && rl < cur.rows_valid;
        const int t = row_ok ? cur.row0 + rl : 0;
        const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)delta, 0, (RES ? 2 : 1) * T * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)rowsum, 0, (RES ? 2 : 1) * L * T * 4, 0x00020000);
        const int tsel = rpass ? T : 0;
        const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (tsel + t) * 4, 0, 0));
        int rsi[L];
#pragma unroll
        for (int l = 0; l < L; ++l) rsi[l] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (L * tsel + l * T + t) * 4, 0, 0);
        const int d2bits = RES ? __builtin_amdgcn_raw_buffer_load_b32(rsD, (T + t) * 4, 0, 0) : 0;
        const bool addp = RES && (d2bits & 0x7fffffff) != 0;
        issue_sz(nxt);

        const __amdgpu_buffer_rsrc_t rsWc = weight_rsrc(cur.e), rsWn = weight_rsrc(nxt.e);
        const int sWc = w_base(cur), sWn = w_base(nxt);
        const int sAc = a_base(cur), sAn = a_base(nxt);
        int sA0 = sAc;                                       // activation offset of the stage being computed

        using T_ = std::true_type;
        using F_ = std::false_type;
        constexpr int NVA = NVF;
        const int nfr_k = cur.nfr;
        v16i acc[L][NF - NVF > 0 ? NF - NVF : 1];
        v16i accv[L][NVF > 0 ? NVF : 1];

        // ---- one 256-k stage.  FIRST: the accumulators start from the instruction's zero operand.
        auto stage = [&](auto first_tag, int kt) {
            constexpr bool FIRST = decltype(first_tag)::value;
            FQL_W4STAMP(ev++, 0);
            const char *sb = lds + (fs & 1) * C::W_STAGE;
            char *nb = lds + ((fs + 1) & 1) * C::W_STAGE;
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
            int sAq[2][L];
#pragma unroll
            for (int l = 0; l < L; ++l) {
                sAq[0][l] = sA0 == OOB ? OOB : sA0 + l * a_limb;
                sAq[1][l] = sA1 == OOB ? OOB : sA1 + l * a_limb;
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
                int rFo = rF0;
                asm volatile("" : "+v"(rFo));
                const char *fc = sb + (rFo ^ frag_xor(ks));                                 // this step's fragments
                const char *fp = ((ks == KS - 1) ? nb : sb) + (rFo ^ frag_xor((ks + 1) & 7));   // the coming step's
                const bool park = ks < NF && !(W4_ABLATE & 1);
                uint32_t up[8];
                auto mm = [&](int j, int l) {                // one matrix instruction of fragment j < NF - NVF
                    if (FIRST && ks == 0) {
                        const v16i z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], afr[ks % D][l], z, 0, 0, 0);
                    } else {
                        acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[j], afr[ks % D][l], acc[l][j], 0, 0, 0);
                    }
                };
                auto fence = [&]() { __builtin_amdgcn_sched_barrier(0); };
                auto refill = [&](int l) {
                    // the ring slot the PREVIOUS step consumed gets the k-step D - 1 ahead of this one (of this stage, or of
                    // the one after it)
                    if (W4_ABLATE & 2) return;
                    const int pk = (ks + KS - 1) % KS, tk = ks - 1 + D, w = tk < KS ? 0 : 1;
                    afr[pk % D][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, sAq[w][l] + (tk % KS) * 1024, 0);
                };
                auto frag = [&](const char *base, int j) { if (!(W4_ABLATE & 4)) wf[j] = *reinterpret_cast<const v4i *>(base + j * 8192); };
                auto unpack = [&](int i) {
                    if (!park) return;
                    unpack8((uint32_t)bst[ks % NF][i], up[2 * i], up[2 * i + 1]);
                    asm volatile("" : "+v"(up[2 * i]), "+v"(up[2 * i + 1]));              // pin the unpack to this slot
                };
                static_assert(NF == 6 && L == 3 && NVF == 1, "the slot plan below is written for 6 fragments x 3 limbs, the last fragment in VGPRs");
                mm(0, 0); fence(); frag(fc, 5); fence();
                mm(0, 1); fence(); frag(fc, 4); fence();
                mm(0, 2); fence(); refill(0); fence();
                if (ks == KS - 1) {
                    // every wave has parked the next stage (steps 0..NF-1) and holds the last fragments of this one
                    wait_lgkmcnt0();
                    __builtin_amdgcn_s_barrier();
                }
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
                mm(4, 1); fence(); if (park) *reinterpret_cast<v4i *>(nb + wA0 + ks * 8192) = v4i{(int)up[0], (int)up[1], (int)up[2], (int)up[3]}; fence();
                mm(4, 2); fence(); if (park) *reinterpret_cast<v4i *>(nb + wA1 + ks * 8192) = v4i{(int)up[4], (int)up[5], (int)up[6], (int)up[7]}; fence();
                if ((W4_ABLATE & 8) || NF - 1 < nfr_k) {     // (wave-uniform: the tile may be narrower than this fragment)
#pragma unroll
                    for (int l = 0; l < L; ++l) {
                        if (FIRST && ks == 0) mfma_i8_vgpr_zero(accv[l][0], wf[NF - 1], afr[ks % D][l]);
                        else mfma_i8_vgpr(accv[l][0], wf[NF - 1], afr[ks % D][l]);
                    }
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
            if (NVA > 0 && j >= NF - NVF) return accv[l][j - (NF - NVF) < 0 ? 0 : j - (NF - NVF)][r];
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
        const int nfr_c = cur.nfr < NF ? cur.nfr : NF;
        auto out4 = [&](int j, int q, const v4f &s4, const v4f &z4, float (&o)[4]) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float tot = 0.0f;
#pragma unroll
                for (int l = L - 1; l >= 0; --l)
                    tot = fmaf(tot, 256.0f, fmaf(-z4[c], rs[l], (float)acc_val(l, j, 4 * q + c)));
                o[c] = (tot * d) * s4[c];
            }
        };
        const bool fast = mode == 0 && vec && out_kind == 0 && bias == nullptr && cur.n0 + nfr_c * 32 <= N;
        if (row_ok_e && !(W4_ABLATE & 16)) {
            // One fragment (32 columns = 16 outputs of this lane) at a time: all its arithmetic as straight-line code, then
            // ONE branch on how to store -- the common case (float32 outputs, whole 16-byte stores, every column inside N)
            // or the general one.  (Two copies of the arithmetic, one per case, get their accumulator reads hoisted in
            // front of the branch by the compiler -- all 288 registers at once.)
            float *orow = reinterpret_cast<float *>(out) + (size_t)t_e * N + cur.n0 + 4 * g_e;
            float *slot0 = res_scratch + ((size_t)blockIdx.x * C::NW + wave) * (NF * 1024) + lane_e * 4;
            // (the scale / zero-point vectors of fragment j + 1 are read from LDS while fragment j is computed: read where
            //  they are used, each of the 48 reads is followed by a full LDS round trip with nothing else to issue)
            v4f sq[2][4], zq[2][4];
            auto read_sz = [&](int j) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = j * 32 + 8 * q + 4 * g_e;
                    sq[j & 1][q] = *reinterpret_cast<const v4f *>(sz + c0);
                    zq[j & 1][q] = *reinterpret_cast<const v4f *>(sz + C::BN + c0);
                }
            };
            read_sz(0);
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                __builtin_amdgcn_sched_barrier(0);           // (register pressure: no fragment's reads before its turn)
                if (j >= nfr_c) continue;
                if (j + 1 < NF) read_sz(j + 1);
                __builtin_amdgcn_sched_barrier(0);
                float o[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) out4(j, q, sq[j & 1][q], zq[j & 1][q], o[q]);
                if (fast) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<v4f *>(orow + j * 32 + 8 * q) = v4f{o[q][0], o[q][1], o[q][2], o[q][3]};
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c0 = j * 32 + 8 * q + 4 * g_e;
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
                            store_out4(out, out_kind, (size_t)t_e * N, cur.n0 + c0, N, vec, o[q]);
                        }
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        FQL_W4STAMP(ev++, 0);
        FQL_W4STAMP(ev++, 1);
        if (!nxt.nt) break;                                  // past the last tile of this workgroup
        cur = nxt;
        parity ^= 1;
    }
#endif  // __HIP_DEVICE_COMPILE__
}
