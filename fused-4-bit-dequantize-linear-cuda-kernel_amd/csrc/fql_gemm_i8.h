// Grouped INT4-weight x INT8-limb-activation GEMM on the CDNA4 matrix cores
// (v_mfma_i32_32x32x32_i8), one launch for all experts.  The linear op is the 1-group case.
//
//   out[t][n] = scale[e][n] * delta[t] * sum_l 256^l * ( sum_k q[e][n][k] * a_l[t][k]  -  zp[e][n] * rowsum_l[t] )
//
// for every row t of expert e's range.  The inner integer dot products are exact (i32
// accumulation), so results do not depend on tile shape, K order or which GPU ran the row.
//
// One workgroup = 8 waves (WM x WN; 4 or 2 for the skinny tiles), tile BM = 32*WM rows x BN = 32*NF*WN columns,
// persistent launch (one 8-wave workgroup per CU walks its tiles):
//   * packed weights (the HBM stream, read once): 8 rows x 128 B = 8 FULL cache lines per wave
//     instruction, global -> VGPR ring (BD stages) -> LDS (two LDS stages, XOR-swizzled so the
//     fragment reads are bank-conflict-free); read back with one ds_read_b128 per 32 columns x 64 k;
//     nibbles are unpacked in registers with 3 VALU ops per 8 weights (unpack8) straight into the MFMA
//     operand, one k-step ahead of the MFMAs that use them.
//   * activation limbs (re-read per column tile, served by the XCD's L2): the pre-pass stores them in
//     MFMA-fragment order, so a wave's A operand for one 32-deep k-step is ONE coalesced 1 KiB
//     buffer_load_dwordx4 straight into VGPRs -- no LDS round trip, no LDS-DMA (whose ~25 B/clk/CU
//     ceiling and ~100-cycle issue cost capped the earlier LDS-staged version); a D-step register ring
//     keeps D k-steps of A in flight per wave (2 at 3 limbs -- the register budget -- up to a full stage of
//     8 at 1 or 2 limbs, where it removes the waits behind the in-order HBM loads).
//   * the weights are the MFMA's A operand, so every lane owns one output row: 16-byte stores; the next
//     tile's first weight stage and its scale / zero-point slice (through LDS) are issued before the
//     current tile's epilogue.
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
    static constexpr int LDS_BYTES = 2 * B_STAGE + 2 * 3 * BN * 4;   // two weight stages + two scale / zero-point / bias slices
    static constexpr int CPWB = BN / 8 / NW;                 // 1 KiB weight pieces per wave per stage
    static_assert(NW == 8 || NW == 4 || NW == 2, "8 waves (two per SIMD), or small 4 / 2-wave workgroups for skinny tiles");
    static_assert(KS % D == 0, "ring depth must divide the steps per stage");
    static_assert((BN / 8) % NW == 0, "weight pieces must divide evenly over the waves");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
    static constexpr int SZN = (3 * BN + THREADS - 1) / THREADS;      // scale / zero-point / bias floats staged per thread
};

// Debug builds (-DFQL_TRACE, tools/trace_kernel.py): wave 0 of the first 8 workgroups stamps the shader clock
// (s_memtime) at every phase boundary and the constant 100 MHz clock (s_memrealtime) at tile boundaries.
#if defined(FQL_TRACE)
__device__ unsigned long long fql_trace_wide[8 * 64];
#define FQL_WSTAMP(i, real) do { if (blockIdx.x < 8 && threadIdx.x == 0 && (i) < 64) fql_trace_wide[blockIdx.x * 64 + (i)] = (real) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define FQL_WSTAMP(i, real) do { } while (0)
#endif

#ifndef FQL_WIDE_W_AUX
#define FQL_WIDE_W_AUX 0       // cache policy of the wide kernel's weight loads (experiment hook; 2 = nt)
#endif

struct GemmTile {              // wave-uniform description of one visit of a BM x BN tile
    int e, row0, prow0, rows_valid, nt, ok;
    int n0, nfr;               // wide kernel: first column and 32-column fragments of this tile (tiles of one row block
                               // differ by at most one fragment: see tile_params)
    int rp;                    // this visit is the RESIDUAL pass of a tile with heavy-tailed rows (its main pass follows)
    int ad;                    // this visit is the main pass that follows a residual pass: add the parked partial results
};

// Heavy-tailed rows (csrc/fql_act_quant.h, pass 3): rows whose 8L-1 fixed-point bits would not carry the stated
// precision get a second limb set for the rounding residual, stored behind the first one -- limbs2 = limbs + L planes,
// delta2 = delta + T (0 for rows without one), rowsum2 = rowsum + L*T.  A tile with such a row is visited twice by
// its workgroup: the residual pass parks float32 partial results delta2 * scale * (...) in a workgroup-private
// scratch slot, the main pass that follows adds them (rows with delta2 == 0 skip the add, so they are bit-identical
// to a tile without the residual pass: a row's result never depends on which other rows share its tile).
// The probe is split so that its L2 round trip hides under other work: `issue` right after the tile is known (two
// unconditional 4-byte loads per lane), `eval` (a ballot) where the answer is first needed.
struct ResidualProbe { int v0, v1; };
__device__ __forceinline__ ResidualProbe residual_probe_issue(const float *delta, int T, const GemmTile &tp, int bm, int lane,
                                                              bool enable)
{
    // always issued (a load under a branch costs hipcc's counted waits); disabled = out-of-bounds offsets = zeros
    const __amdgpu_buffer_rsrc_t rsD2 = __builtin_amdgcn_make_buffer_rsrc((void *)(delta + T), 0, T * 4, 0x00020000);
    const bool on = enable && tp.ok;
    ResidualProbe p;
    p.v0 = __builtin_amdgcn_raw_buffer_load_b32(rsD2, (on && lane < tp.rows_valid) ? (tp.row0 + lane) * 4 : 0x7fff0000, 0, 0);
    p.v1 = __builtin_amdgcn_raw_buffer_load_b32(rsD2, (on && bm > 64 && lane + 64 < tp.rows_valid) ? (tp.row0 + lane + 64) * 4 : 0x7fff0000, 0, 0);
    return p;
}
__device__ __forceinline__ int residual_probe_eval(const ResidualProbe &p)
{
    return __builtin_amdgcn_readfirstlane((__ballot(((p.v0 | p.v1) & 0x7fffffff) != 0) != 0ull) ? 1 : 0);
}
__device__ __forceinline__ int tile_has_residual(const float *delta, int T, const GemmTile &tp, int bm, int lane)
{
    return residual_probe_eval(residual_probe_issue(delta, T, tp, bm, lane, true));
}

__device__ __forceinline__ void wait_lgkmcnt0() { __builtin_amdgcn_s_waitcnt(15 | (7 << 4) | (0 << 8) | (3 << 14)); }


// F8 (L = 1 only): the activation bytes are OCP e4m3 values (csrc/fql_act_f8.h) instead of int8 limbs and the
// contraction runs on v_mfma_scale_f32_32x32x64_f8f6f4 -- one instruction per PAIR of 32-deep k-steps, float32
// accumulation.  The weight operand needs no conversion: the nibble byte 0000qqqq read as e4m3 (exponent field 0
// or 1) IS q * 2^-9, and the instruction's E8M0 block scale of the weight operand is set to 2^9, so the
// accumulator holds sum_k q * a.  Which true k a byte slot holds is irrelevant to the matrix core as long as both
// operands agree, so the fragment layouts of the INT8 path are used unchanged (checked on hardware:
// tools/micro/mfma_f8_probe.hip).
template <int L, int WM, int WN, int NF, int DEPTH, int BDEPTH, bool F8 = false>
__global__ __launch_bounds__(64 * WM * WN, 2) void gemm_i8_kernel(
    const int8_t *__restrict__ limbs, const float *__restrict__ delta,
    const int32_t *__restrict__ rowsum, const uint8_t *__restrict__ packed,
    const float *__restrict__ scales, const float *__restrict__ zps, void *__restrict__ out, int out_kind_flags,
    const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int K, int Kp, int MBT, int N, int n_tiles_min, int m_slots, float *__restrict__ res_scratch,
    const float *__restrict__ bias, int n_tiles_alt)
{
#if defined(__HIP_DEVICE_COMPILE__)   // the host pass only needs the launch stub
    using C = GemmCfg<L, WM, WN, NF, DEPTH, BDEPTH>;
    constexpr int KS = C::KS, D = C::D;
    // out_kind_flags: bits 0-1 the output element type (FQL_DTYPE_*), bit 3: multiply every output row by its row weight
    // (plane delta[sets * T + t], written by the pre-pass of fql_moe_gather_scaled_fwd_f32: the routing weight folded into
    // the epilogue, so that the combine step is a pure gather-add).  One rounding, after the bias: (x W^T + b) * w.
    const int out_kind = out_kind_flags & 3;
    const bool row_scaled = (out_kind_flags & 8) != 0;
    constexpr bool RES = FQL_RES_ENABLED && (L >= 2) && !F8;                    // residual limb set for heavy-tailed rows (see GemmTile)
    static_assert(!F8 || (L == 1 && D % 2 == 0), "the fp8 form has one activation byte plane and consumes k-steps in pairs");
    using acc_t = typename std::conditional<F8, v16f, v16i>::type;
    extern __shared__ __attribute__((aligned(16))) char lds[];

    // ---- tiles.  Real m-tiles are counted on the device (expert counts live there).  The launch is
    //      PERSISTENT: one workgroup per CU walks virtual block ids vb = blockIdx.x, +gridDim.x, ...
    //      Logical tile ids are m-tile major and dealt to XCDs in contiguous ranges (vb % 8 = the XCD
    //      group of the workgroup, for every vb it visits), so the workgroups of one XCD share an
    //      expert's activation panel in that XCD's L2 while each weight byte streams once.
    //      The NEXT tile's first loads (its stage-0 weights, its scale / zero-point slice, its first
    //      activation fragments) are issued BEFORE the current tile's epilogue arithmetic and stores, so the
    //      HBM round trip of one tile's prologue hides under the other's VALU-bound epilogue.
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, g = lane >> 5;
    int n_tiles = n_tiles_min;
    int n_real = m_slots * n_tiles;
    // ---- columns: the N / 32 fragments of a row block are dealt to its n_tiles tiles as evenly as possible, the wider
    //      tiles first: tile i holds base + (i < rem) fragments.  With the fewest tiles that cover N this is the plain
    //      BN-wide tiling; the host may ask for more, narrower tiles so that the tiles every persistent workgroup walks
    //      add up alike (8 x 128 rows x 11008 columns: 58 tiles of 192 per block = 464 tiles = 2 rounds for 1.81;
    //      64 tiles of 192 / 160 = 512 tiles, every workgroup one wide and one narrow tile).
    //      Which of the two tile counts the host offers (the fewest, or its balanced alternative) is decided HERE, from
    //      the row-block count the device-side expert counts actually give: rounds of the busiest workgroup x (average
    //      fragments per tile + 1 for prologue / epilogue) -- the host's guess assumes even routing, and a count that evens
    //      out 8 row blocks is a round too many for 13.
    const int n_frag = (N + 31) >> 5;
    auto pick_tiles = [&](int m_tiles) {
        if (n_tiles_alt <= 0) return;
        const int G = (int)gridDim.x;
        const float ra = (float)((m_tiles * n_tiles_min + G - 1) / G), rb = (float)((m_tiles * n_tiles_alt + G - 1) / G);
        const float ca = ra * (float)(n_frag + n_tiles_min) * (float)n_tiles_alt;      // r (F / t + 1), cross-multiplied
        const float cb = rb * (float)(n_frag + n_tiles_alt) * (float)n_tiles_min;
        if (cb < ca) n_tiles = n_tiles_alt;
    };
    if (tpe != nullptr) {
        // One vector load per 64 experts (every wave does it redundantly; nothing is shared).
        int cp = 0, ct = 0;
        for (int base = 0; base < E; base += 64) (void)expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
        const int m_tiles = __builtin_amdgcn_readfirstlane(ct < m_slots ? ct : m_slots);     // overlapping ranges: stay inside the plan
        pick_tiles(m_tiles);
        n_real = m_tiles * n_tiles;
    } else {
        pick_tiles(m_slots);
        n_real = m_slots * n_tiles;
    }
    n_tiles = __builtin_amdgcn_readfirstlane(n_tiles);
    n_real = __builtin_amdgcn_readfirstlane(n_real);
    const int f_base = n_frag / n_tiles, f_rem = n_frag - f_base * n_tiles;

    auto tile_params = [&](int vb, int lane) -> GemmTile {   // wave-uniform (lane passed in: see the note at the epilogue)
        GemmTile tp = {0, 0, 0, 0, 0, 0, 0, 0};
        if (vb >= n_real) return tp;
        const int tile = xcd_remap(vb, n_real);
        const int ms = tile / n_tiles;
        tp.nt = tile - ms * n_tiles;
        tp.nfr = f_base + (tp.nt < f_rem ? 1 : 0);
        tp.n0 = (tp.nt * f_base + (tp.nt < f_rem ? tp.nt : f_rem)) * 32;
        if (tpe == nullptr) {                                // linear: one group covering all T rows
            tp.row0 = tp.prow0 = ms * C::BM;
            tp.rows_valid = T - tp.row0;
            tp.ok = 1;
        } else {                                             // MoE: the expert that owns this m-tile
            int cp = 0, ct = 0;
            for (int base = 0; base < E && !tp.ok; base += 64) {
                const ExpertLane x = expert_chunk(tpe, offs, E, T, C::BM, base, lane, cp, ct);
                const unsigned long long hit = __ballot(ms >= x.tile_excl && ms < x.tile_excl + x.tiles);
                if (hit) {
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
    static_assert(C::BM <= 128, "the residual probe covers two 64-row halves");

    // ---- everything per-lane is tile independent; the tile enters through scalar offsets and descriptors
    constexpr int OOB = 0x7fff0000;                          // a buffer offset past every descriptor: reads zero
    constexpr int BD = C::BD;
    const int KB = Kp / FQL_KB, KT = KB;                     // weight stages (K padded to 256 by the pre-pass)
    const size_t wbytes = (size_t)N * (size_t)(K >> 1);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void *)limbs, 0, (int)((size_t)(RES ? 2 : 1) * L * KB * MBT * 8192), 0x00020000);
    const int a_stage = MBT * 8192;                          // bytes between consecutive kb of one limb
    // A operand: limbs[l][kb][mb][ks][lane][16 B]
    // (one base register per address family; the fragment / piece / limb index goes into the scalar offset or the
    //  instruction's immediate, which keeps ~10 VGPRs out of a 256-register kernel)
    const int aoff0 = lane * 16;
    const int a_limb = KB * MBT * 8192;                      // bytes between the limbs of one (kb, mb) block
    // weight staging: piece p = i*8 + wave covers rows 8p..8p+7 of the tile, 128 B each (8 full lines)
    //  piece i: rows + i * 8 * NW -> + i * 8 * NW * (K/2) bytes in memory, + i * NW KiB in the LDS image (the
    //  XOR swizzle repeats every 16 rows); fragment j: rows + 32 j -> + 4 KiB j in LDS, same swizzle
    const int row0B = wave * 8 + (lane >> 3), chB = lane & 7;
    const int voffB0 = row0B * (K >> 1) + chB * 16;
    const int pieceB = 8 * C::NW * (K >> 1);
    const int wB0 = row0B * 128 + 16 * (chB ^ ((row0B >> 1) & 7));      // swizzled LDS image
    const int swB0 = (l31 >> 1) & 7;                                     // (32-row steps do not move the swizzle)
    // scale / zero-point slice of a tile: 2 * BN floats through LDS (thread i < BN: scale of column i, thread
    // BN + i: zero point), double buffered by tile parity so the epilogue needs no global loads for them
    float *szbuf = reinterpret_cast<float *>(lds + 2 * C::B_STAGE);

    v4i bst[BD][C::CPWB];                                    // weight stages in flight (global -> VGPR -> LDS)
    v4i afr[D][L];                                           // A ring: D k-steps ahead
    float szr[C::SZN];

    // first loads of a tile.  Every load is UNCONDITIONAL (a missing tile, rows past N and columns past N read
    // zero through the descriptors): a load under an `if` makes hipcc's counted vmcnt collapse to "wait for
    // almost everything", which throws the prefetch lead away.
    // scalar offset of weight piece i (rows 8 (i NW + wave) .. + 7 of the tile): past the tile's own fragments the rows
    // belong to the next tile -- out of bounds, so they read zero and cost no traffic
    auto piece_off = [&](int so, int i, int nfr) -> int { return (8 * (i * C::NW + wave) < nfr * 32) ? so + i * pieceB : OOB; };
    auto issue_prologue = [&](const GemmTile &tp, int tid) {   // (tid passed in: see the note at the epilogue)
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(packed + (size_t)tp.e * wbytes), 0, (int)wbytes, 0x00020000);
        const int sB = tp.ok ? tp.n0 * (K >> 1) : OOB;
#pragma unroll
        for (int i = 0; i < C::CPWB; ++i) bst[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voffB0, piece_off(sB, i, tp.nfr), FQL_WIDE_W_AUX);
        const __amdgpu_buffer_rsrc_t rsS = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(scales + (size_t)tp.e * N), 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsZ = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(zps + (size_t)tp.e * N), 0, N * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsBi = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(bias != nullptr ? bias + (size_t)tp.e * N : scales), 0, bias != nullptr ? N * 4 : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < C::SZN; ++i) {
            const int idx = tid + i * C::THREADS;            // < BN: scale of column idx; < 2 BN: zero point; then the bias
            const int arr = idx < C::BN ? 0 : (idx < 2 * C::BN ? 1 : 2);
            const int col = idx - arr * C::BN;
            const int so = (tp.ok && idx < 3 * C::BN) ? 0 : OOB;
            const int vo = (tp.n0 + col) * 4;
            const int vs = __builtin_amdgcn_raw_buffer_load_b32(rsS, arr == 0 ? vo : OOB, so, 0);
            const int vz = __builtin_amdgcn_raw_buffer_load_b32(rsZ, arr == 1 ? vo : OOB, so, 0);
            const int vb = __builtin_amdgcn_raw_buffer_load_b32(rsBi, arr == 2 ? vo : OOB, so, 0);
            szr[i] = __builtin_bit_cast(float, vs | vz | vb);   // the other two read zero (no bias: a zero-length descriptor)
        }
    };

  int ev = 0; (void)ev;               // (only the FQL_TRACE build reads it)
  GemmTile cur = tile_params(blockIdx.x, lane);
  if constexpr (RES) {
      const ResidualProbe pb = residual_probe_issue(delta, T, cur, C::BM, lane, res_scratch != nullptr);
      issue_prologue(cur, tid);
      cur.rp = residual_probe_eval(pb);
  } else {
      issue_prologue(cur, tid);
  }
  int parity = 0;
  for (int vb = blockIdx.x; vb < n_real; parity ^= 1) {      // one iteration per VISIT (tile, pass); vb advances below
    FQL_WSTAMP(ev++, 1);                                     // tile start, constant 100 MHz clock
    FQL_WSTAMP(ev++, 0);                                     // tile start, shader clock
    const int e = cur.e, row0 = cur.row0, rows_valid = cur.rows_valid;
    const int n0 = cur.n0, nfr = cur.nfr;
    // Which wave takes which 32-row block (wm) and column half (wn) is decided per tile.  Wave w runs on SIMD w % 4, so
    //   wm = w % WM, wn = w / WM puts the column halves of one row block on one SIMD: a tile that is one fragment
    //     narrower (tile_params) takes that fragment's MFMAs off every SIMD alike (a full tile: -3 %);
    //   wm = w / WN, wn = w % WN puts two different row blocks on a SIMD: when at most half of the row blocks hold rows
    //     (a short group inside a 128-row tile) every SIMD keeps one working wave instead of half the SIMDs two (-14 %).
    const bool spread_rows = rows_valid <= (WM / 2) * FQL_MB;
    const int wm = spread_rows ? wave / WN : wave % WM, wn = spread_rows ? wave % WN : wave / WM;
    const int rB0 = (wn * NF * 32 + l31) * 128;
    const int nfw = nfr - wn * NF < 0 ? 0 : (nfr - wn * NF > NF ? NF : nfr - wn * NF);   // this wave's fragments that exist
    const bool rpass = RES && cur.rp != 0;                   // residual pass: second limb set, partials to the scratch slot
    const bool active = cur.ok && wm * FQL_MB < rows_valid; // waves past the expert's last row only help stage weights
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(packed + (size_t)e * wbytes), 0, (int)wbytes, 0x00020000);
    const int sB = n0 * (K >> 1);                            // scalar part of this tile's weight offsets
    const int sA = ((cur.prow0 >> 5) + wm) * 8192 + (rpass ? L * a_limb : 0);   // ... and of its activation offsets
    float *sz = szbuf + parity * 3 * C::BN;

    acc_t acc[L][NF];
#pragma unroll
    for (int l = 0; l < L; ++l)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[l][j][r] = 0;

   if (cur.ok) {
    // ---- stage 0 of the weights (in flight since the previous tile's epilogue) into LDS, stages 1..BD into
    //      the staging-register ring (slot of stage s = s % BD)
#pragma unroll
    for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(lds + wB0 + i * C::NW * 1024) = bst[0][i];
#pragma unroll
    for (int i = 0; i < C::SZN; ++i)
        if (tid + i * C::THREADS < 3 * C::BN) sz[tid + i * C::THREADS] = szr[i];
#pragma unroll
    for (int s = 1; s <= BD; ++s)
#pragma unroll
        for (int i = 0; i < C::CPWB; ++i)
            bst[s % BD][i] = __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB0, piece_off(s < KT ? sB + s * (FQL_KB / 2) : OOB, i, nfr), FQL_WIDE_W_AUX);

    // The K loop exists in two forms: all NF fragments of this wave, or NF - 1 when the tile is one fragment narrower
    // and the missing fragment is this wave's last (tile_params: tiles of a row block differ by at most one fragment).
    // Fewer fragments still (the ragged last tile of N, shapes narrower than a tile) run the full form on zeros, as ever.
    auto k_loop = [&](auto nfa_tag) {
        constexpr int NFA = decltype(nfa_tag)::value;
        // ---- the weight fragments are software-pipelined one k-step ahead: the ds_reads of the next 64-k
        //      pair and the nibble unpack of the next step are issued under the current step's MFMAs.  One
        //      barrier per stage, placed at step 5: by then every wave has parked stage kt+1 (its step 0) and
        //      has finished reading stage kt (the last read of it is issued at step 4).
        //      (The activation ring starts here, not with the early prologue: its 8 * L * D registers would be
        //      live across the previous tile's epilogue, and these are L2 hits.)
#pragma unroll
        for (int s = 0; s < D; ++s)
#pragma unroll
            for (int l = 0; l < L; ++l) afr[s][l] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aoff0, sA + l * a_limb + s * 1024, 0);
        wait_lgkmcnt0();
        __builtin_amdgcn_s_barrier();
        // ping-pong register sets indexed by compile-time parity (the k-step loop is fully unrolled), so the
        // hand-over from "next" to "current" costs no register moves
        v4i bfr2[2][NF], braw2[2][NF];
#pragma unroll
        for (int j = 0; j < NFA; ++j) {
            braw2[0][j] = *reinterpret_cast<const v4i *>(lds + rB0 + j * 4096 + 16 * ((0 + g) ^ swB0));
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
                    // then refill that ring slot with stage kt+1+BD (BD stages of HBM lead; past the last stage
                    // the offset is out of bounds: zeros, no memory traffic).
                    const int sNext = (kt + 1 + BD < KT) ? sB + (kt + 1 + BD) * (FQL_KB / 2) : OOB;
                #pragma unroll
                    for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(nb + wB0 + i * C::NW * 1024) = bst[(kk + 1) % BD][i];
#pragma unroll
                    for (int i = 0; i < C::CPWB; ++i)
                        bst[(kk + 1) % BD][i] =
                            __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB0, piece_off(sNext, i, nfr), FQL_WIDE_W_AUX);
                }
                if (ks == 5) {
                    wait_lgkmcnt0();
                    __builtin_amdgcn_s_barrier();
                }
                if (b == 0) {                  // raw weights of the NEXT pair (pair 0 of the next stage at step 6)
                    const char *src = (v < 3) ? sb : nb;
                    const int nv = (v + 1) & 3;
#pragma unroll
                    for (int j = 0; j < NFA; ++j)
                        braw2[pn][j] = *reinterpret_cast<const v4i *>(src + rB0 + j * 4096 + 16 * ((2 * nv + g) ^ swB0));
                }
#if defined(FQL_ABLATE) && FQL_ABLATE == 1
#pragma unroll
                for (int l = 0; l < L; ++l) asm volatile("" ::"v"(afr[ks % D][l]));
#pragma unroll
                for (int j = 0; j < NFA; ++j) asm volatile("" ::"v"(bfr2[b][j]));
#else
                if constexpr (F8) {
                    if (b == 1) {              // steps ks-1 and ks together: 64 k per instruction
                        const v8i a8 = {afr[(ks - 1) % D][0][0], afr[(ks - 1) % D][0][1], afr[(ks - 1) % D][0][2], afr[(ks - 1) % D][0][3],
                                        afr[ks % D][0][0], afr[ks % D][0][1], afr[ks % D][0][2], afr[ks % D][0][3]};
#pragma unroll
                        for (int j = 0; j < NFA; ++j) {
                            const v8i w8 = {bfr2[0][j][0], bfr2[0][j][1], bfr2[0][j][2], bfr2[0][j][3],
                                            bfr2[1][j][0], bfr2[1][j][1], bfr2[1][j][2], bfr2[1][j][3]};
                            acc[0][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8, a8, acc[0][j], 0, 0, 0, FQL_E8M0_2P9, 0, FQL_E8M0_ONE);
                        }
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < L; ++l)
#pragma unroll
                        for (int j = 0; j < NFA; ++j)
                            acc[l][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bfr2[b][j], afr[ks % D][l], acc[l][j], 0, 0, 0);
                }
#endif
#pragma unroll
                for (int j = 0; j < NFA; ++j) {  // unpack for the next step under the MFMAs
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
                if constexpr (F8) {
                    if (b == 1) {
#pragma unroll
                        for (int s = ks - 1; s <= ks; ++s)
                            afr[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(
                                rsA, aoff0, sA + (kt + (s + D) / KS) * a_stage + ((s + D) % KS) * 1024, 0);
                    }
                } else {
#pragma unroll
                    for (int l = 0; l < L; ++l)
                        afr[ks % D][l] = __builtin_amdgcn_raw_buffer_load_b128(
                            rsA, aoff0, sA + l * a_limb + (kt + nks / KS) * a_stage + (nks % KS) * 1024, 0);
                }
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
    };
    if (active) {
        if constexpr (NF >= 2 && !F8) {
            if (nfw == NF - 1) k_loop(std::integral_constant<int, NF - 1>{});
            else k_loop(std::integral_constant<int, NF>{});
        } else {
            k_loop(std::integral_constant<int, NF>{});
        }
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
            const int sNext = (kt + 1 + BD < KT) ? sB + (kt + 1 + BD) * (FQL_KB / 2) : OOB;
#pragma unroll
            for (int i = 0; i < C::CPWB; ++i) *reinterpret_cast<v4i *>(nb + wB0 + i * C::NW * 1024) = bst[(kk + 1) % BD][i];
#pragma unroll
            for (int i = 0; i < C::CPWB; ++i)
                bst[(kk + 1) % BD][i] =
                    __builtin_amdgcn_raw_buffer_load_b128(rsB, voffB0, piece_off(sNext, i, nfr), FQL_WIDE_W_AUX);
            wait_lgkmcnt0();
            __builtin_amdgcn_s_barrier();
          }
        }
    }
   }   // cur.ok

    FQL_WSTAMP(ev++, 0);                                     // K loop done
    // the next visit: the main pass of this tile after its residual pass, else the next tile (its expert-table
    // loads are short and nothing slow is ahead of them in the load queue)
    GemmTile nxt;
    ResidualProbe pb = {0, 0};
    // (per-lane values from here to the end of the visit are derived from an opaque copy of the thread id: derived from
    //  the kernel-entry copies they are invariants of the persistent tile loop, live across the K loop, and spilled around it)
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    const int lane_e = tid_e & 63, l31_e = lane_e & 31, g_e = lane_e >> 5;
    if (rpass) { nxt = cur; nxt.rp = 0; nxt.ad = 1; }
    else { vb += (int)gridDim.x; nxt = tile_params(vb, lane_e); }
    if constexpr (RES) pb = residual_probe_issue(delta, T, nxt, C::BM, lane_e, res_scratch != nullptr && !rpass);   // evaluated after the epilogue
    // ---- epilogue: fold zero-point, combine limbs, scale.  The weights are the MFMA's A operand (rows = n)
    //      and the activations its B operand (cols = t), so in the 32x32 C/D layout
    //      (col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)) every lane owns ONE output
    //      row t and registers 4q..4q+3 are 4 consecutive output columns: 4 per-row loads per lane, the
    //      scale / zero-point vectors from LDS, 16-byte stores.
    //      Order of issue matters (vector-memory loads complete in order): the four short per-row loads first,
    //      then the next tile's long HBM loads, then the arithmetic and the stores.
    const int rl = wm * FQL_MB + l31_e;
    const bool row_ok = active && rl < rows_valid;
    const int t = row_ok ? row0 + rl : 0;
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)delta, 0, ((RES ? 2 : 1) + 1) * T * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void *)rowsum, 0, (RES ? 2 : 1) * L * T * 4, 0x00020000);
    const int tsel = rpass ? T : 0;                          // second set of per-row values in the residual pass
    const float d = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, (tsel + t) * 4, 0, 0));
    int rsi[L];
#pragma unroll
    for (int l = 0; l < L; ++l) rsi[l] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (L * tsel + l * T + t) * 4, 0, 0);
    // main pass after a residual pass: does THIS row have a residual (delta2 != 0)?
    const int d2bits = RES ? __builtin_amdgcn_raw_buffer_load_b32(rsD, (T + t) * 4, 0, 0) : 0;   // unconditional load
    const bool addp = RES && (d2bits & 0x7fffffff) != 0;
    const float rw = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsD, row_scaled ? ((RES ? 2 : 1) * T + t) * 4 : OOB, 0, 0));
    __builtin_amdgcn_sched_barrier(0);
    issue_prologue(nxt, tid_e);
    __builtin_amdgcn_sched_barrier(0);
    // MODE 0: plain tile (the hot path: straight-line code, nothing of the residual machinery in it);
    // MODE 1: residual pass -- park the float32 results in this lane's scratch slot (workgroup-private; read back by
    //         the same lane in the next visit);  MODE 2: main pass after a residual pass -- add the parked values.
    auto epilogue = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;
        float rs[L];
#pragma unroll
        for (int l = 0; l < L; ++l) rs[l] = (float)rsi[l];
        const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & (out_kind == 0 ? 15 : 7)) == 0);
        // (the lane part goes through an opaque register: otherwise the compiler hoists all NF * 4 slot pointers out of
        //  the persistent tile loop, keeps them live across the K loop and spills them -- 72 bytes of scratch per lane)
        const int lane4 = lane_e * 4, g4 = 4 * g_e;
        float *slot0 = (MODE == 0) ? nullptr
                                   : res_scratch + ((size_t)blockIdx.x * C::NW + wave) * (NF * 1024) + lane4;
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (j >= nfw) continue;                                      // columns of the next tile
                const int c0 = (wn * NF + j) * 32 + 8 * q + g4;             // column inside the tile
                const v4f s4 = *reinterpret_cast<const v4f *>(sz + c0);
                const v4f z4 = *reinterpret_cast<const v4f *>(sz + C::BN + c0);
                float o[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float tot = 0.0f;
                    if constexpr (F8) {
                        tot = fmaf(-z4[c], __builtin_bit_cast(float, rsi[0]), acc[0][j][4 * q + c]);
                    } else {
#pragma unroll
                        for (int l = L - 1; l >= 0; --l)
                            tot = fmaf(tot, 256.0f, fmaf(-z4[c], rs[l], (float)acc[l][j][4 * q + c]));
                    }
                    o[c] = (tot * d) * s4[c];
                }
                if constexpr (MODE == 1) {
                    *reinterpret_cast<v4f *>(slot0 + (j * 4 + q) * 256) = v4f{o[0], o[1], o[2], o[3]};
                } else {
                    if constexpr (MODE == 2) {
                        if (addp) {                          // rows without a residual stay bit-identical to MODE 0
                            const v4f pr = *reinterpret_cast<const v4f *>(slot0 + (j * 4 + q) * 256);
#pragma unroll
                            for (int c = 0; c < 4; ++c) o[c] += pr[c];
                        }
                    }
                    if (bias != nullptr) {                   // (uniform; without a bias the results keep their exact bits)
                        const v4f b4 = *reinterpret_cast<const v4f *>(sz + 2 * C::BN + c0);
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] += b4[c];
                    }
                    if (row_scaled) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) o[c] *= rw;
                    }
                    store_out4(out, out_kind, (size_t)t * N, n0 + c0, N, vec, o);
                }
            }
    };
    if (row_ok) {
        if (!RES || (!rpass && cur.ad == 0)) epilogue(std::integral_constant<int, 0>{});
        else if (rpass) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, 2>{});
    }
    if constexpr (RES) { if (!rpass) nxt.rp = residual_probe_eval(pb); }
    cur = nxt;
    FQL_WSTAMP(ev++, 0);                                     // epilogue issued
    FQL_WSTAMP(ev++, 1);
  }   // persistent tile loop
#endif  // __HIP_DEVICE_COMPILE__
}
