// Grouped INT4 x INT8-limb GEMM for DECODE-SIZE row groups (<= 16 rows per expert per tile): the op is a weight
// stream (4.5 flop per weight byte per row), so the kernel is organised around one rule learnt from the phase
// traces of the 32-row kernel: vector-memory loads of a wave complete IN ORDER, so a
// short-lead load issued after a long-lead (HBM) load waits for the whole HBM round trip.  Here EVERY load of a
// wave has the same lead of one full 256-k stage and is issued at a stage boundary:
//   * weights of the next stage(s): 8 rows x 128 B full lines -> VGPR ring (BD stages) -> wave-private LDS slab,
//   * ALL activation fragments of the next stage: v_mfma_i32_16x16x64_i8 needs only 16 rows, so a whole stage of
//     activations is 4 steps x L limbs x 4 registers (48 at 3 limbs) -- it fits in registers twice (current +
//     next), which the 32x32x32 shape could not afford,
//   * the tile's scale / zero-point slice and per-row values with its first stage (scales through LDS),
//   * the next tile's first stages while the current tile finishes (rings run across tile boundaries).
// Inside a stage a wave issues only LDS reads, unpacks and MFMAs.  No barrier in the K loop: the 8 waves of a
// workgroup split K KG ways and the columns 8/KG ways and own their weight rows privately; the KG partial sums
// are added pairwise through LDS at the end (int32: exact).  Bit-identical to every other tile configuration.
//
// Activation fragments come from the same fragment-native limb workspace as the 32-row kernels: the 16-byte
// piece (k-step ks, lane group g, row r) of a 256-k block holds k = 64 v + 32 g + 16 b (+0..15) with ks = 2v + b;
// the 64-k MFMA step s takes k-slice kq = lane >> 4 from piece (ks = 2s + (kq & 1), g = kq >> 1), and the weight
// operand takes the same slice from bytes [8 (kq & 1), +8) of 16-byte chunk 2s + (kq >> 1) of its row's segment.
#pragma once
#include "fql_common.h"
#include "fql_gemm_i8.h"

#ifndef FQL_W_AUX
#define FQL_W_AUX 2            // cache policy of the weight-stream loads: nt -- every byte is read once, by one wave
#endif                         // (measured at 8 experts x 8 rows: 51.3 -> 47.4-48.8 us against the default policy)

template <int L, int NF, int KG, int BDEPTH, int NWAVES = 8>
struct Rows16Cfg {
    static constexpr int NW = NWAVES;                         // 8: one workgroup per CU; 4: TWO independent workgroups per CU (one's
                                                              // reduction / epilogue / start under the other's weight stream)
    static constexpr int NG = NW / KG;
    static constexpr int THREADS = 64 * NW;
    static constexpr int BM = 16;
    static constexpr int BN = 16 * NF * NG;
    static constexpr int BD = BDEPTH;
    static constexpr int UNR = (BDEPTH % 2 == 0) ? BDEPTH : 2 * BDEPTH;   // stages per unrolled trip: ring slot and
                                                                          // activation parity are compile-time
    static constexpr int PIECES = NF * 2;                     // 1 KiB weight pieces (8 rows x 128 B) per wave per stage
    static constexpr int SLAB = NF * 16 * (FQL_KB / 2);       // wave-private LDS bytes: one stage of packed weights
    static constexpr int ACC_BYTES = L * NF * 4 * 64 * 4;
    static constexpr int SZ_BYTES = 2 * 3 * BN * 4;           // scale / zero-point / bias slices of two tiles
    // K-group partial sums: FLAT = every wave parks its accumulators once and fragment j of a column group is summed
    // and finished (epilogue, store) by the wave with kg = j % KG -- one exchange instead of a log2(KG)-round tree
    // and the epilogue spread over the K groups; the tree remains for shapes whose partials do not fit in LDS
    static constexpr bool FLAT = (KG > 1) && (NW * ACC_BYTES + SZ_BYTES <= 160 * 1024);
    static constexpr int RED_BYTES = (KG > 1) ? (FLAT ? NW * ACC_BYTES : (KG / 2) * NG * ACC_BYTES) : 0;
    static constexpr int MAIN_BYTES = (NW * SLAB > RED_BYTES) ? NW * SLAB : RED_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES + SZ_BYTES;
    static constexpr int SZN = (3 * BN + THREADS - 1) / THREADS;
    static_assert(KG == 1 || KG == 2 || KG == 4 || KG == 8, "K split");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int L, int NF, int KG, int BDEPTH, int NWAVES = 8>
__global__ __launch_bounds__(64 * NWAVES, 2) void gemm_i8_rows16_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, void *__restrict__ out, int out_kind_flags,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles_min, int m_slots, float *__restrict__ res_scratch,
    const float *__restrict__ bias, int n_tiles_alt)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Rows16Cfg<L, NF, KG, BDEPTH, NWAVES>;
    // out_kind_flags: bits 0-1 the output element type (FQL_DTYPE_*), bit 3: multiply every output row by its row weight
    // (plane delta[sets * T + t], written by the pre-pass of fql_moe_gather_scaled_fwd_f32: the routing weight folded into
    // the epilogue, so that the combine step is a pure gather-add).  One rounding, after the bias: (x W^T + b) * w.
    const int out_kind = out_kind_flags & 3;
    const bool row_scaled = (out_kind_flags & 8) != 0;
    constexpr bool RES = FQL_RES_ENABLED && (L >= 2);                           // residual limb set for heavy-tailed rows (fql_gemm_i8.h)
    constexpr int NG = C::NG, BD = C::BD;
    constexpr int OOB = 0x7fff0000;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kg = wave / NG, ng = wave - kg * NG;
    const int l15 = lane & 15, kq = lane >> 4;

    // ---- expert table: ONE pass over the device-side counts per workgroup, kept in LDS (first 64 experts; more
    //      experts fall back to re-reading per tile), so finding a tile's expert later costs no global load and no
    //      wait behind the weight stream
    __shared__ int s_lo[64], s_cnt[64], s_tex[64], s_pex[64];
    // ---- columns: the N / 16 fragments of a row block are dealt evenly to its tiles, the wider tiles first, and a
    //      tile's fragments evenly to its NG column groups (this is a weight stream: what a workgroup walks is the bytes
    //      it reads).  Of the two tile counts the host offers -- the fewest that cover N and a balanced alternative --
    //      the one whose busiest workgroup walks less is picked here, from the row-block count the device-side expert
    //      counts give (8 experts x <= 16 rows x 11008 columns: 86 tiles of 128 per block = 688 tiles = 3 rounds for
    //      2.69; 96 tiles of 8 / 7 fragments = 768 tiles, three per workgroup).
    const int n_frag = (N + 15) >> 4;
    int n_tiles = n_tiles_min;
    auto pick_tiles = [&](int m_tiles) {
        if (n_tiles_alt <= 0) return;
        const int G = (int)gridDim.x;
        const float ra = (float)((m_tiles * n_tiles_min + G - 1) / G), rb = (float)((m_tiles * n_tiles_alt + G - 1) / G);
        const float ca = ra * (float)(n_frag + 2 * n_tiles_min) * (float)n_tiles_alt;     // r (F / t + 2), cross-multiplied
        const float cb = rb * (float)(n_frag + 2 * n_tiles_alt) * (float)n_tiles_min;
        if (cb < ca) n_tiles = n_tiles_alt;
    };
    int n_real;
    if (tpe != nullptr) {
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) {
            const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
            if (base == 0 && wave == 0) { s_lo[lane] = x.lo; s_cnt[lane] = x.cnt; s_tex[lane] = x.tile_excl; s_pex[lane] = x.pad_excl; }
        }
        const int m_tiles = __builtin_amdgcn_readfirstlane(ct < m_slots ? ct : m_slots);
        pick_tiles(m_tiles);
        n_real = m_tiles * n_tiles;
    } else {
        pick_tiles(m_slots);
        n_real = m_slots * n_tiles;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    n_real = __builtin_amdgcn_readfirstlane(n_real);
    const int f_base = n_frag / n_tiles, f_rem = n_frag - f_base * n_tiles;
    __syncthreads();

    auto tile_params = [&](int vb) -> GemmTile {
        GemmTile tp = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vb >= n_real) return tp;
        const int tile = xcd_remap(vb, n_real);
        const int ms = tile / n_tiles;
        tp.nt = tile - ms * n_tiles;
        tp.nfr = f_base + (tp.nt < f_rem ? 1 : 0);
        tp.n0 = (tp.nt * f_base + (tp.nt < f_rem ? tp.nt : f_rem)) * 16;
        if (tpe == nullptr) {
            tp.row0 = tp.prow0 = ms * C::BM;
            tp.rows_valid = T - tp.row0;
            tp.ok = 1;
        } else {
            // first 64 experts from LDS (lane i looks at expert i)
            {
                const int lo = s_lo[lane], cnt = s_cnt[lane], te = s_tex[lane], pe = s_pex[lane];
                const int tiles_e = (cnt + C::BM - 1) / C::BM;
                const unsigned long long hit = __ballot(lane < E && ms >= te && ms < te + tiles_e);
                if (hit) {
                    const int src = __ffsll((long long)hit) - 1;
                    const int lo_s = wave_bcast(lo, src), cnt_s = wave_bcast(cnt, src);
                    const int te_s = wave_bcast(te, src), pe_s = wave_bcast(pe, src);
                    tp.e = src;
                    tp.row0 = lo_s + (ms - te_s) * C::BM;
                    tp.prow0 = pe_s + (ms - te_s) * C::BM;    // experts are padded to 32 rows: a 16-row tile is one half block
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
        tp.n0 = __builtin_amdgcn_readfirstlane(tp.n0);
        tp.nfr = __builtin_amdgcn_readfirstlane(tp.nfr);
        tp.ok = __builtin_amdgcn_readfirstlane(tp.ok);
        return tp;
    };
    // this wave's share of a tile's fragments: column group ng takes fragments [ng * nf0, ng * nf0 + nfw) of the tile
    auto frag0 = [&](const GemmTile &tp) -> int { return ng * ((tp.nfr + NG - 1) / NG); };
    auto frags = [&](const GemmTile &tp) -> int {
        const int nf0 = (tp.nfr + NG - 1) / NG, left = tp.nfr - ng * nf0;
        return left < 0 ? 0 : (left > nf0 ? nf0 : left);
    };

    const int KB = Kp / FQL_KB, KT = KB;
    const int SP = ((KT + KG - 1) / KG + C::UNR - 1) / C::UNR * C::UNR;   // stages per wave per tile, padded (even)
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)(RES ? 2 : 1) * L * KB * MBT * 8192), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)delta, 0, ((RES ? 2 : 1) + 1) * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)rowsum, 0, (RES ? 2 : 1) * L * T * 4, 0x00020000);
    const int a_stage = MBT * 8192;
    const int a_limb = KB * MBT * 8192;

    // ---- tile-independent per-lane offsets
    const int aoff0 = ((kq & 1) * 64 + (kq >> 1) * 32 + l15) * 16;   // piece (ks = 2s + (kq&1), g = kq>>1, row) of a block
    char *slab = lds + wave * C::SLAB;
    const int rowW = lane >> 3, chW = lane & 7;                // piece i: rows 8i + rowW
    const int vrel0 = rowW * (K >> 1) + chW * 16;
    // LDS image: row * 128 + 16 * (chunk ^ ((row >> 1) & 7)); row = 8 i + rowW, so the swizzle term alternates
    // with the parity of the piece
    const int wB0 = rowW * 128 + 16 * (chW ^ ((rowW >> 1) & 7));            // even pieces (+ i * 1024)
    const int wB1 = rowW * 128 + 16 * (chW ^ (((rowW >> 1) + 4) & 7));      // odd pieces
    const int rB0 = l15 * 128 + 8 * (kq & 1);                  // fragment j: + j * 2048; chunk 2s + (kq>>1), swizzled
    const int swB0 = (l15 >> 1) & 7;
    float *szbuf = reinterpret_cast<float *>(lds + C::MAIN_BYTES);

    auto w_rsrc = [&](const GemmTile &tp) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)(packed + (size_t)tp.e * wbytes), 0, (int)wbytes, 0x00020000);
    };
    auto w_soff = [&](const GemmTile &tp, int s) -> int {      // stage s of this K group in tile tp, or out of bounds
        const int kt = kg + s * KG;
        return (tp.ok && kt < KT) ? (tp.n0 + frag0(tp) * 16) * (K >> 1) + kt * (FQL_KB / 2) : OOB;
    };
    auto a_soff = [&](const GemmTile &tp, int s) -> int {
        const int kt = kg + s * KG;
        return (tp.ok && kt < KT) ? (tp.prow0 >> 5) * 8192 + ((tp.prow0 >> 4) & 1) * 256 + kt * a_stage + (tp.rp ? L * a_limb : 0) : 0;
    };

    // ---- rings: BD weight stages + ONE full activation stage ahead, running across tile boundaries
    v4i bst[BD][C::PIECES];
    v4i afr[2][4][L];                                          // [parity of the stage][64-k step][limb]
    float szr[C::SZN];
    int drow[1 + L];                                           // delta bits and limb row sums of this lane's row
    int rwbits = 0;                                           // this lane's row weight (row_scaled)
    int d2bits = 0;                                            // delta2 bits of this lane's row (heavy-tailed rows: fql_gemm_i8.h)
    auto issue_weights = [&](const __amdgpu_buffer_rsrc_t rs, int so, int slot, int nfw) {   // pieces past this wave's fragments: no traffic
#pragma unroll
        for (int i = 0; i < C::PIECES; ++i)
            bst[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, vrel0, (8 * i < 16 * nfw) ? so + i * 8 * (K >> 1) : OOB, FQL_W_AUX);
    };
    auto issue_acts = [&](int so, int par) {
#if defined(FQL_ABLATE) && FQL_ABLATE == 2          // timing experiment only (wrong results): no activation traffic
        if (so != 0x12345678) return;
#endif
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int l = 0; l < L; ++l)
                afr[par][s][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, so + l * a_limb + s * 2048, 0);
    };
    auto issue_tile_consts = [&](const GemmTile &tp) {         // scale / zero-point slice and this lane's row values
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
            const int vo = (tp.n0 + col) * 4;
            const int vs = __builtin_amdgcn_raw_buffer_load_b32(rsS, arr == 0 ? vo : OOB, so, 0);
            const int vz = __builtin_amdgcn_raw_buffer_load_b32(rsZ, arr == 1 ? vo : OOB, so, 0);
            const int vb = __builtin_amdgcn_raw_buffer_load_b32(rsBi, arr == 2 ? vo : OOB, so, 0);
            szr[i] = __builtin_bit_cast(float, vs | vz | vb);
        }
        const int t = (tp.ok && l15 < tp.rows_valid) ? tp.row0 + l15 : 0;
        const int tsel = tp.rp ? T : 0;                        // residual pass: the second set of per-row values
        drow[0] = __builtin_amdgcn_raw_buffer_load_b32(rsD, (tsel + t) * 4, 0, 0);
#pragma unroll
        for (int l = 0; l < L; ++l) drow[1 + l] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (L * tsel + l * T + t) * 4, 0, 0);
        if (RES) d2bits = __builtin_amdgcn_raw_buffer_load_b32(rsD, (T + t) * 4, 0, 0);
        if (row_scaled) rwbits = __builtin_amdgcn_raw_buffer_load_b32(rsD, ((RES ? 2 : 1) * T + t) * 4, 0, 0);
    };

    int ev = 0; (void)ev;
    FQL_STAMP(ev++);                                           // kernel entry
    GemmTile cur = tile_params(blockIdx.x);
    {
        // first tile: every load goes out at once -- weights, the activations of the MAIN pass and the tile constants;
        // the heavy-tail answer for the tile's rows is read meanwhile through the SCALAR cache (uniform addresses, its
        // own counter: it does not queue behind the weight stream the way a vector load would), and only a tile that
        // does have such rows re-issues its activations and constants for the residual pass
        const __amdgpu_buffer_rsrc_t rs = w_rsrc(cur);
#pragma unroll
        for (int u = 0; u < BD; ++u) issue_weights(rs, w_soff(cur, u), u, frags(cur));
        issue_acts(a_soff(cur, 0), 0);
        issue_tile_consts(cur);
        if constexpr (RES) {
            if (res_scratch != nullptr && cur.ok) {
                const uint32_t *d2 = reinterpret_cast<const uint32_t *>(delta) + T + cur.row0;
                const int last = cur.rows_valid - 1;
                uint32_t v[C::BM];
#pragma unroll
                for (int i = 0; i < C::BM; ++i) v[i] = d2[i < last ? i : last];
                uint32_t any = 0;
#pragma unroll
                for (int i = 0; i < C::BM; ++i) any |= v[i];
                cur.rp = __builtin_amdgcn_readfirstlane((any & 0x7fffffffu) != 0 ? 1 : 0);
                if (cur.rp) {
                    issue_acts(a_soff(cur, 0), 0);
                    issue_tile_consts(cur);
                }
            }
        }
    }
    int parity = 0;                                            // tile parity (scale slice buffer)

  for (int vb = blockIdx.x; vb < n_real; parity ^= 1) {      // one iteration per visit (tile, pass): see GemmTile
    FQL_STAMP(ev++);                                           // tile start
    GemmTile nxt;
    const bool rpass = RES && cur.rp != 0;
    if (rpass) { nxt = cur; nxt.rp = 0; nxt.ad = 1; }
    else { vb += gridDim.x; nxt = tile_params(vb); }
    // heavy-tail probe of the next tile: issued now, evaluated at the last stage boundary (where nxt is first used)
    ResidualProbe pb = {0, 0};
    if constexpr (RES) pb = residual_probe_issue(delta, T, nxt, C::BM, lane, res_scratch != nullptr && !rpass);
    FQL_STAMP(ev++);                                           // next tile known
    const __amdgpu_buffer_rsrc_t rs_cur = w_rsrc(cur), rs_nxt = w_rsrc(nxt);
    const int nfw_cur = frags(cur), nfw_nxt = frags(nxt), fr0_cur = frag0(cur);
    float *sz = szbuf + parity * 3 * C::BN;
    // this tile's constants arrived with its first stage: park the scale slice, keep the row values
#pragma unroll
    for (int i = 0; i < C::SZN; ++i)
        if (tid + i * C::THREADS < 3 * C::BN) sz[tid + i * C::THREADS] = szr[i];
    const float d = __builtin_bit_cast(float, drow[0]);
    const float rw = __builtin_bit_cast(float, rwbits);
    const bool addp = RES && (d2bits & 0x7fffffff) != 0;
    float rsum[L];
#pragma unroll
    for (int l = 0; l < L; ++l) rsum[l] = (float)drow[1 + l];

    v4i acc[L][NF];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int j = 0; j < NF; ++j) acc[l][j] = v4i{0, 0, 0, 0};

    for (int s0 = 0; s0 < SP; s0 += C::UNR) {
#pragma unroll
      for (int uu = 0; uu < C::UNR; ++uu) {
        const int s = s0 + uu;
        constexpr int dummy = 0; (void)dummy;
        const int u = uu % BD;                                 // weight ring slot (static after unrolling)
        const int apar = uu & 1;                               // activation stage parity (SP is even: static)
        FQL_STAMP(ev++);                                       // stage start
        // ---- stage boundary: park this stage's weights, then issue EVERYTHING the next stage(s) need
#pragma unroll
        for (int i = 0; i < C::PIECES; ++i) *reinterpret_cast<v4i *>(slab + ((i & 1) ? wB1 : wB0) + i * 1024) = bst[u][i];
        FQL_STAMP_FINE(ev++);                                  // this stage's weights have arrived and are parked
        {   // order matters (loads complete in order): the L2-resident activations of the NEXT stage first, the HBM
            // weights of the stage BD ahead last, so the next boundary's wait for the activations does not include
            // the youngest HBM loads
            if constexpr (RES) { if (s + 1 == SP && !rpass) nxt.rp = residual_probe_eval(pb); }
            const bool ahere = s + 1 < SP;
            issue_acts(ahere ? a_soff(cur, s + 1) : a_soff(nxt, 0), apar ^ 1);
            if (s + 1 == SP) issue_tile_consts(nxt);          // with the next tile's first activation stage
            const bool here = s + BD < SP;
            issue_weights(here ? rs_cur : rs_nxt, here ? w_soff(cur, s + BD) : w_soff(nxt, s + BD - SP), u, here ? nfw_cur : nfw_nxt);
        }
        FQL_STAMP_FINE(ev++);                                  // next loads issued
        // ---- 4 MFMA steps of 64 k: LDS reads, unpack, matrix cores; no memory instruction in here
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            v4i wfr[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const uint2 raw = *reinterpret_cast<const uint2 *>(slab + rB0 + j * 2048 + 16 * ((2 * st + (kq >> 1)) ^ swB0));
                uint32_t lo0, hi0, lo1, hi1;
                unpack8(raw.x, lo0, hi0);
                unpack8(raw.y, lo1, hi1);
                wfr[j] = v4i{(int)lo0, (int)hi0, (int)lo1, (int)hi1};
            }
#pragma unroll
            for (int l = 0; l < L; ++l)
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    acc[l][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfr[j], afr[apar][st][l], acc[l][j], 0, 0, 0);
            if (st == 0) FQL_STAMP_FINE(ev++);                  // first step's operands were there (activations arrived)
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    FQL_STAMP(ev++);                                           // K loop done
    // ---- add the KG partial accumulators through LDS (int32: exact, order-free)
    if constexpr (C::FLAT) {
        __syncthreads();                                       // every wave is done with its weight slab
#pragma unroll
        for (int l = 0; l < L; ++l)
#pragma unroll
            for (int j = 0; j < NF; ++j)
                *reinterpret_cast<v4i *>(lds + wave * C::ACC_BYTES + ((l * NF + j) * 64 + lane) * 16) = acc[l][j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (j % KG != kg) continue;                        // fragment j belongs to the wave with kg = j % KG
#pragma unroll
            for (int l = 0; l < L; ++l) {
                v4i tot = v4i{0, 0, 0, 0};
#pragma unroll
                for (int g2 = 0; g2 < KG; ++g2)
                    tot += *reinterpret_cast<const v4i *>(lds + (g2 * NG + ng) * C::ACC_BYTES + ((l * NF + j) * 64 + lane) * 16);
                acc[l][j] = tot;
            }
        }
        __syncthreads();                                       // the slabs are free again for the next tile's weights
    } else if (KG > 1) {
#pragma unroll
        for (int sft = 1; sft < KG; sft <<= 1) {
            char *red = lds + ((kg / (2 * sft)) * NG + ng) * C::ACC_BYTES;
            __syncthreads();
            if ((kg & (2 * sft - 1)) == sft) {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
                        *reinterpret_cast<v4i *>(red + ((l * NF + j) * 64 + lane) * 16) = acc[l][j];
            }
            __syncthreads();
            if ((kg & (2 * sft - 1)) == 0) {
#pragma unroll
                for (int l = 0; l < L; ++l)
#pragma unroll
                    for (int j = 0; j < NF; ++j) {
                        const v4i p = *reinterpret_cast<const v4i *>(red + ((l * NF + j) * 64 + lane) * 16);
                        acc[l][j] += p;
                    }
            }
        }
        __syncthreads();
    } else {
        __syncthreads();                                       // the scale slice parked by other waves
    }

    // ---- epilogue: D = weights x activations, so lane (t = lane & 15, nq = lane >> 4) owns row t and registers
    //      0..3 are columns 4 nq .. 4 nq + 3 of each 16-column fragment: one 16-byte store per fragment.
    //      Scales / zero points from LDS, the row values from registers: no global load.
    FQL_STAMP(ev++);                                           // reduction done
    const GemmTile done = cur;
    cur = nxt;
    if ((!C::FLAT && kg != 0) || !done.ok || l15 >= done.rows_valid) continue;
    const int t = done.row0 + l15;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & (out_kind == 0 ? 15 : 7)) == 0);
    // MODE 0 plain tile / 1 residual pass (park float32 results in the scratch slot) / 2 main pass after it (add them)
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        float *slot0 = (MODE == 0) ? nullptr : res_scratch + ((size_t)blockIdx.x * C::NW + wave) * (NF * 256) + lane * 4;
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            if (C::FLAT && j % KG != kg) continue;              // finished by the wave that summed it
            if (j >= nfw_cur) continue;                         // columns of the next tile
            const int c0 = (fr0_cur + j) * 16 + 4 * kq;           // column inside the tile
            const v4f s4 = *reinterpret_cast<const v4f *>(sz + c0);
            const v4f z4 = *reinterpret_cast<const v4f *>(sz + C::BN + c0);
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float tot = 0.0f;
#pragma unroll
                for (int l = L - 1; l >= 0; --l)
                    tot = fmaf(tot, 256.0f, fmaf(-z4[c], rsum[l], (float)acc[l][j][c]));
                o[c] = (tot * d) * s4[c];
            }
            if constexpr (MODE == 1) {
                *reinterpret_cast<v4f *>(slot0 + j * 256) = v4f{o[0], o[1], o[2], o[3]};
            } else {
                if constexpr (MODE == 2) {
                    if (addp) {
                        const v4f pr = *reinterpret_cast<const v4f *>(slot0 + j * 256);
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
                store_out4(out, out_kind, (size_t)t * N, done.n0 + c0, N, vec, o);
            }
        }
    };
    if (!RES || (done.rp == 0 && done.ad == 0)) epilogue(std::integral_constant<int, 0>{});
    else if (done.rp) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 2>{});
  }
#endif
}
