// libfql_int4.so -- C ABI (include/fql_int4.h) over the gfx950 kernels.
//
// Host-side dispatch only: argument validation, kernel selection, launches on the caller's
// stream.  No allocation, no synchronisation, no state.
#include "../../include/fql_int4.h"
#include "../../include/fql_int4_tune.h"
#include "fql_common.h"
#include "fql_act_quant.h"
#include "fql_act_f8.h"
#include "fql_gemm_i8.h"
#include "fql_gemm_rows32.h"
#include "fql_gemm_rows16.h"
#include "fql_gemv.h"
#include "fql_group.h"
#include "fql_group_i8.h"
#include "fql_generic.h"
#include "fql_quantize.h"
#include "fql_routing.h"
#include "fql_w4_launch.h"
#include <atomic>
#include <random>

namespace {

// Batches up to this many rows take the float32 GEMV kernel, larger ones the MFMA path (measured on MI355X, 4096 -> 11008,
// product call in a hipGraph: GEMV 9.3 / 12.1 / 19.7 us at B = 1 / 2 / 3, MFMA path 16.4 / 16.5 / 16.1 us at
// B = 2 / 3 / 4 and 16.2-16.5 us at B = 5..16: profiles/r02_linear_batch_sweep.txt).  Tuning hook below.
int g_gemv_max_rows = 2;
int g_group_i8 = 1;                    // per-group scales: the INT8 matrix-core kernel where eligible (A/B hook below)
int g_group_i8_min_rows = 40;          // ... from this many rows on for one matrix (below: the float32 matrix-core kernel; measured
int g_group_i8_min_rows_grouped = 8;   //     crossover 32..48), and from 8 rows per expert on for grouped calls (190 vs 230 us at 8 x 8 rows)
int g_group_mfma = 1;                  // per-group scales: the float32 matrix-core kernel for batches (A/B hook below)
int g_use_w4 = 1;                      // 3 limbs, > 64 rows per group: the one-wave-per-SIMD kernel (fql_gemm_w4.h) instead of the 8-wave 128 x 192 one (A/B hook below)
int g_act_single_rows = 512;          // pre-pass: one row per workgroup up to this many padded rows (tuning hook below)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline size_t round16(size_t v) { return (v + 15) & ~(size_t)15; }
inline int padded_k(int K) { return (K + FQL_KB - 1) / FQL_KB * FQL_KB; }
// 32-row blocks of the limb workspace: every expert starts on a block boundary (<= 31 pad rows each) and a
// tile may run up to 128 rows past the last expert.
inline int row_blocks(int T, int E) { return (T + FQL_MB * E + 128 + FQL_MB - 1) / FQL_MB; }

inline int limbs_of(int precision)
{
    if (precision == FQL_PRECISION_DEFAULT) return 3;
    if (precision == FQL_PRECISION_INT8 || precision == FQL_PRECISION_FAST || precision == FQL_PRECISION_EXACT)
        return precision;
    if (precision == FQL_PRECISION_FP8) return 1;            // one byte plane of e4m3 values
    return -1;
}
inline bool is_f8(int precision) { return precision == FQL_PRECISION_FP8; }

// Per-device caches (a process may drive several GPUs): indexed by the current device, idempotent -- a race only repeats
// the same query / attribute call.
constexpr int FQL_MAX_DEVICES = 64;
inline int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    return dev < FQL_MAX_DEVICES ? dev : FQL_MAX_DEVICES - 1;
}
struct PerDeviceFlag {
    bool set[FQL_MAX_DEVICES] = {};
};
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and device
inline bool ensure_lds_attr(PerDeviceFlag &flag, const void *kern, int bytes)
{
    const int dev = current_device();
    if (!flag.set[dev]) {
        if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
        flag.set[dev] = true;
    }
    return true;
}
int g_cu_cap = 0;              // tuning hook (tests): pretend the device has this many compute units (multiple of 8; 0 = real count)
inline int compute_units()
{
    static int cached[FQL_MAX_DEVICES] = {};
    if (g_cu_cap > 0) return g_cu_cap;
    const int dev = current_device();
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;                                         // MI355X
        n -= n % 8;                                          // keep vb % 8 == blockIdx % 8 (XCD grouping)
        cached[dev] = n > 0 ? n : 8;
    }
    return cached[dev];
}

inline int compute_units_hint() { const int n = compute_units(); return n < 256 ? 256 : n; }

struct Workspace {            // (every member has a default: a hand-filled Workspace must not carry garbage pointers into a launch)
    int8_t *limbs = nullptr;
    float *delta = nullptr;
    int32_t *rowsum = nullptr;
    float *scratch = nullptr;          // workgroup-private float32 partials of the residual pass (heavy-tailed rows), or nullptr
    const float *bias = nullptr;       // optional per-column bias [N] added to the final outputs (not workspace memory: rides along)
    const float *row_weight = nullptr; // optional per-row output weight [T] (caller's array: the pre-pass copies it into the plane behind delta)
    unsigned long long *flags = nullptr;   // one-launch form (fql_gemm_w4.h, FUSED): one word per group of 4 grouped rows
    const void *fuse_x = nullptr;      // one-launch form requested: the float32 rows the GEMM kernel quantises itself (ride along, like bias)
    const int32_t *fuse_gather = nullptr;
    int fuse_n_src = 0;
    size_t bytes = 0;
};

// Modes with a residual limb set for heavy-tailed rows (csrc/fql_act_quant.h pass 3, csrc/fql_gemm_i8.h): 2 and 3 limbs.
inline bool has_residual(int L, bool f8) { return L >= 2 && !f8; }
// Upper bound of the residual-pass scratch over every tile configuration: workgroups x tile floats
// (8 waves x <= 4 fragments x 4 KiB per workgroup and CU; the small skinny-tile workgroups share a CU's budget).
inline size_t res_scratch_bytes() { return (size_t)compute_units_hint() * 8 * 4 * 4096; }

inline size_t limb_bytes(int L, int T, int E, int Kp) { return (size_t)L * (Kp / FQL_KB) * row_blocks(T, E) * 8192; }

inline Workspace carve(void *base, int L, int T, int E, int Kp, bool res)
{
    Workspace w;
    const int sets = res ? 2 : 1;
    const size_t lb = round16(sets * limb_bytes(L, T, E, Kp));
    const size_t db = round16((size_t)(sets + 1) * T * sizeof(float));       // + the row-weight plane (fql_moe_gather_scaled_fwd_f32)
    const size_t rb = round16((size_t)sets * L * T * sizeof(int32_t));
    const size_t fb = (L == 3 && res) ? round16(((size_t)T + 3) / 4 * sizeof(unsigned long long)) : 0;   // row-group flags of the one-launch form
    char *p = static_cast<char *>(base);
    w.limbs = reinterpret_cast<int8_t *>(p);
    w.delta = reinterpret_cast<float *>(p + lb);
    w.rowsum = reinterpret_cast<int32_t *>(p + lb + db);
    w.flags = fb ? reinterpret_cast<unsigned long long *>(p + lb + db + rb) : nullptr;
    w.scratch = res ? reinterpret_cast<float *>(p + lb + db + rb + fb) : nullptr;
    w.bias = nullptr;
    w.row_weight = nullptr;
    w.bytes = lb + db + rb + fb + (res ? res_scratch_bytes() : 0);
    return w;
}

// ---- MFMA tile configurations (see fql_gemm_i8.h): 8 waves as WM x WN, NF 32-column fragments per wave.
struct TileShape { int bm, bn; };
// X(id, WM, WN, NF, A-ring depth in k-steps, weight stages in flight)
#define FQL_CFG_LIST(X)                                                                                            \
    X(0, 4, 2, 3, 2, 1)        /* 128 x 192 (3 limbs: 2-step A ring is what the register budget allows) */ \
    X(1, 4, 2, 2, 4, 1)        /* 128 x 128 */ \
    X(2, 4, 2, 4, 2, 1)        /* 128 x 256 (2-limb register budget) */ \
    X(3, 4, 2, 3, 4, 1)        /* 128 x 192, 4-step A ring (2-limb register budget) */ \
    X(4, 2, 4, 2, 4, 1)        /*  64 x 256 */ \
    X(5, 1, 8, 1, 4, 1)        /*  32 x 256 */ \
    X(6, 4, 1, 2, 8, 4)        /* 128 x  64, 4 waves: skinny tiles for few rows (HBM-bound: many small */ \
    X(7, 2, 2, 1, 8, 4)        /*  64 x  64, 4 waves   workgroups per CU, 4 weight stages in flight, deep */ \
    X(8, 1, 2, 1, 8, 4)        /*  32 x  64, 2 waves   A ring to cover L2 latency) */ \
    X(9, 2, 4, 3, 2, 1)        /*  64 x 384: 64-row groups with the A-fragment reuse of the 128 x 192 tile */ \
    X(10, 2, 4, 3, 4, 2)       /*  64 x 384, 2 weight stages in flight (1- and 2-limb register budgets) */ \
    X(11, 2, 4, 3, 8, 2)       /*  64 x 384, full-stage A ring (every load one stage ahead), 2 weight stages */ \
    X(12, 4, 2, 3, 8, 2)       /* 128 x 192, full-stage A ring, 2 weight stages */ \
    X(13, 4, 1, 2, 4, 8)       /* 128 x  64, 4 waves, 8 weight stages in flight (few tall tiles: HBM-latency bound) */ \
    X(14, 4, 1, 3, 2, 1)       /* 128 x  96, 4 waves: TWO independent workgroups per CU, one wave per SIMD each, so one */ \
    X(15, 2, 2, 3, 2, 1)       /*  64 x 192, 4 waves   workgroup's prologue / epilogue / barrier waits sit under the other's MFMAs */ \
    X(16, 4, 1, 3, 2, 2)       /* 128 x  96, 4 waves, 2 weight stages in flight */ \
    X(17, 4, 1, 2, 4, 2)       /* 128 x  64, 4 waves, 4-step A ring */
constexpr int FQL_NUM_CFG = 18;
// Short row groups (fql_gemm_rows32.h): 32-row tiles, K split KG ways inside the workgroup.  ids 100 + i.
// R(i, NF, KG, A-ring depth in k-steps, weight stages in flight per wave, waves per SIMD)
#define FQL_ROWS32_LIST(R)                                                                                         \
    R(0, 2, 4, 2, 2, 2)        /* 32 x 128 */ \
    R(1, 2, 8, 2, 1, 2)        /* 32 x  64 */ \
    R(2, 2, 2, 2, 2, 2)        /* 32 x 256 */ \
    R(3, 2, 4, 2, 1, 2)        /* 32 x 128, one weight stage in flight */ \
    R(4, 1, 4, 4, 2, 2)        /* 32 x  64, one fragment per wave */ \
    R(5, 1, 4, 2, 1, 4)        /* 32 x  64, <= 128 registers: two workgroups per CU hide each other's HBM waits */ \
    R(6, 1, 8, 2, 1, 4)        /* 32 x  32, two workgroups per CU */ \
    R(7, 1, 2, 2, 1, 4)        /* 32 x 128, two workgroups per CU */
constexpr int FQL_NUM_ROWS32 = 8;
// Decode-size row groups (fql_gemm_rows16.h): 16-row tiles on v_mfma_i32_16x16x64_i8, every load one stage ahead.
// ids 200 + i.  S(i, NF, KG, weight stages in flight)
#define FQL_ROWS16_LIST(S)                                                                                         \
    S(0, 4, 4, 2)              /* 16 x 128 */ \
    S(1, 4, 2, 2)              /* 16 x 256 */ \
    S(2, 4, 4, 1)              /* 16 x 128, one weight stage in flight */ \
    S(3, 4, 8, 1)              /* 16 x  64 */ \
    S(4, 8, 4, 1)              /* 16 x 256, 8 fragments per wave */ \
    S(5, 4, 8, 2)              /* 16 x  64, both weight stages of a 4096-k row in flight from the start */ \
    S(6, 3, 8, 2)              /* 16 x  48 */ \
    S(7, 3, 8, 1)              /* 16 x  48, one weight stage in flight */ \
    S(8, 2, 8, 2)              /* 16 x  32 */
constexpr int FQL_NUM_ROWS16 = 9;
// (round 3: 16 x 192 and 16 x 160 -- two tiles per workgroup at the decode shape instead of three -- were built and timed: 47.1 / 49.4 us
//  against 46.1 us for id 2, at 253-256 registers; profiles/r03_decode_phase_trace.txt.  Not kept.)
// the same kernel as 4-wave workgroups, two per CU.  ids 220 + i.  S4(i, NF, KG, weight stages in flight)
#define FQL_ROWS16_W4_LIST(S4)                                                                                     \
    S4(0, 4, 4, 1)             /* 16 x  64, K split 4 ways */ \
    S4(1, 4, 4, 2)             /* 16 x  64, two weight stages in flight */ \
    S4(2, 4, 2, 1)             /* 16 x 128, K split 2 ways */ \
    S4(3, 8, 4, 1)             /* 16 x 128, 8 fragments per wave */
constexpr int FQL_NUM_ROWS16_W4 = 4;
// One wave per SIMD (fql_gemm_w4.h, its own translation unit): 128 x 192 tiles, 4 waves of 32 x 192.  ids 300 + i:
// W(i, limbs, fragments per wave, activation ring depth in k-steps)
#define FQL_W4_LIST(W)                                                                                             \
    W(0, 3, 6, 8)              /* 3 limbs, full-stage activation ring */ \
    W(1, 3, 6, 4)              /* 3 limbs, 4-step ring */
constexpr int FQL_NUM_W4 = 2;
// Which wide configurations are BUILT for which limb count: the ones choose_cfg() can return for it, plus the previous
// headline configuration (0 at 3 limbs: the bit-identity baseline of the tests and A/B tools).  The rest of the list --
// register budgets of another limb count (up to 1273 spilled registers), tuning experiments that lost -- is not instantiated.
constexpr bool wide_cfg_built(int L, int id)
{
    return L == 3 ? (id == 0 || id == 1 || id == 7 || id == 8 || id == 9 || id == 13)
         : L == 2 ? (id == 1 || id == 2 || id == 3 || id == 7 || id == 8 || id == 11 || id == 13)
                  : (id == 1 || id == 2 || id == 7 || id == 8 || id == 11 || id == 12 || id == 13);
}
inline bool valid_cfg(int cfg, int L) { return (cfg >= 300 && cfg < 300 + FQL_NUM_W4 && L == 3) || (cfg >= 0 && cfg < FQL_NUM_CFG && wide_cfg_built(L, cfg)) ||
           (cfg >= 100 && cfg < 100 + FQL_NUM_ROWS32) || (cfg >= 200 && cfg < 200 + FQL_NUM_ROWS16) || (cfg >= 220 && cfg < 220 + FQL_NUM_ROWS16_W4);
}

// The MFMA path addresses its operands through 32-bit buffer offsets.
inline bool mfma_addressable(int L, int T, int E, int K, int N, bool f8 = false)
{
    const size_t a = (has_residual(L, f8) ? 2 : 1) * limb_bytes(L, T, E, padded_k(K));   // (with the residual limb set where there is one)
    const size_t b = ((size_t)N + 256) * (size_t)(K >> 1);
    return a < ((size_t)1 << 31) && b < ((size_t)1 << 31);
}

inline bool mfma_eligible(int L, int T, int E, int K, int N, const uint8_t *packed)
{
    return (K % 32 == 0) && aligned16(packed) && mfma_addressable(L, T, E, K, N);
}

inline int dtype_bytes(int dt) { return dt == FQL_DTYPE_F32 ? 4 : 2; }

template <int L>
int launch_act_quant(const void *x, int in_dtype, const int32_t *gather, int n_src, const Workspace &w, int T, int K,
                     int Kp, int MBT, void *out, int out_dtype, int N, const int32_t *tpe, const int32_t *offs, int E,
                     hipStream_t st, bool gated = false, bool f8out = false)
{
    // ACT_ROWS-row workgroups over the grouped rows (each finds its rows' padded positions itself), plus (MoE entry
    // point) the workgroups that zero the rows of `out` no expert covers
    const int mblocks = (tpe == nullptr) ? (T + FQL_MB - 1) / FQL_MB : (T + FQL_MB * E) / FQL_MB;   // (upper bound of the padded row blocks)
    const bool vec = (K % 16 == 0) && (reinterpret_cast<uintptr_t>(x) % 16 == 0);
    // few rows in all (at most two single-row workgroups per CU; measured: 16 rows 6.3 -> 4.6 us, 1280 padded rows 12.5 -> 16.5 us): one row per workgroup -- the pre-pass is a latency
    // chain there and a row spread over 256 threads shortens every link of it (fql_act_quant.h)
    const bool single = vec && mblocks * FQL_MB <= g_act_single_rows;
    const int rblocks = single ? T : (T + ACT_ROWS - 1) / ACT_ROWS;
    const int zblocks = (tpe != nullptr && out != nullptr) ? (T + 255) / 256 : 0;
    void (*kern)(const void *, const int32_t *, int, float *, int32_t *, int8_t *, int, int, int, int, int, void *, int,
                 int, const int32_t *, const int32_t *, int, const float *);
#define FQL_ACT_PICK(l, in, gate, f8) \
    (single ? act_fused_kernel<l, true, in, gate, f8, 1> : (vec ? act_fused_kernel<l, true, in, gate, f8> : act_fused_kernel<l, false, in, gate, f8>))
    if (f8out) {
        if constexpr (L == 1) {
            switch (in_dtype) {
            case FQL_DTYPE_F16: kern = FQL_ACT_PICK(1, 1, false, true); break;
            case FQL_DTYPE_BF16: kern = FQL_ACT_PICK(1, 2, false, true); break;
            default: kern = FQL_ACT_PICK(1, 0, false, true); break;
            }
        } else return FQL_ERR_BAD_PRECISION;
    } else if (gated) kern = FQL_ACT_PICK(L, 0, true, false);
    else switch (in_dtype) {
    case FQL_DTYPE_F16: kern = FQL_ACT_PICK(L, 1, false, false); break;
    case FQL_DTYPE_BF16: kern = FQL_ACT_PICK(L, 2, false, false); break;
    default: kern = FQL_ACT_PICK(L, 0, false, false); break;
    }
#undef FQL_ACT_PICK
    (void)hipGetLastError();
    hipLaunchKernelGGL(kern, dim3(rblocks + zblocks), dim3(256), 0, st, x, gather, n_src, w.delta, w.rowsum, w.limbs,
                       T, K, Kp / FQL_KB, MBT, rblocks, out, dtype_bytes(out_dtype), N, tpe, offs, E, w.row_weight);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

// Column tiles per row block of the wide kernel.  The fewest that cover N (every tile at most `fpt` fragments of 32
// columns) is the plain tiling; a few more, narrower tiles are chosen when that evens out what each persistent
// workgroup walks (fql_gemm_i8.h, tile_params).  Cost model: a tile costs its fragments + 1 (prologue / epilogue),
// the launch costs what its busiest workgroup walks; `m_tiles` is the row-block count under even routing (the real
// count lives on the device).  Exact replay of the kernel's tile order, cached per shape.
int g_balance_tiles = 1;
inline int host_xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}
inline int balanced_n_tiles(int N, int fpt, long long m_tiles, int wgs, int frag = 32, int overhead = 1, int min_base = -1)
{
    if (min_base < 0) min_base = fpt - 1;                    // wide kernel: its K loop has a form for one fragment less than a full tile, not fewer
    const int F = (N + frag - 1) / frag;
    const int tmin = (F + fpt - 1) / fpt;
    if (!g_balance_tiles || m_tiles <= 0 || wgs <= 0 || wgs > 4096 || m_tiles * tmin > 8192) return tmin;
    struct Key { int N, fpt, wgs, frag; long long m_tiles; int result; };
    static thread_local Key cache[8] = {};
    static thread_local int next_slot = 0;
    for (const Key &k : cache)
        if (k.result > 0 && k.N == N && k.fpt == fpt && k.wgs == wgs && k.frag == frag && k.m_tiles == m_tiles) return k.result;
    auto busiest = [&](int t) -> long long {
        const long long n_real = m_tiles * t;
        const int base = F / t, rem = F - base * t;
        const int G = n_real < wgs ? (int)n_real : wgs;
        static thread_local int cost[4096];
        for (int i = 0; i < G; ++i) cost[i] = 0;
        for (int vb = 0; vb < (int)n_real; ++vb) {
            const int i = host_xcd_remap(vb, (int)n_real) % t;
            cost[vb % G] += base + (i < rem ? 1 : 0) + overhead;
        }
        int mx = 0;
        for (int i = 0; i < G; ++i) mx = cost[i] > mx ? cost[i] : mx;
        return mx;
    };
    int best = tmin;
    long long best_cost = busiest(tmin);
    // candidates: tile counts that make the launch a whole number of rounds
    for (long long rounds = (m_tiles * tmin + wgs - 1) / wgs; rounds <= (m_tiles * tmin + wgs - 1) / wgs + 1; ++rounds) {
        if ((rounds * wgs) % m_tiles != 0) continue;
        const long long t = rounds * wgs / m_tiles;
        if (t <= tmin || t > F || F / t < min_base) continue;
        const long long c = busiest((int)t);
        if (c < best_cost) { best_cost = c; best = (int)t; }
    }
    cache[next_slot] = Key{N, fpt, wgs, frag, m_tiles, best};
    next_slot = (next_slot + 1) & 7;
    return best;
}

template <int L, int WM, int WN, int NF, int DEPTH, int BDEPTH, bool F8 = false>
int launch_gemm_cfg(const Workspace &w, const uint8_t *packed, const float *scales, const float *zps,
                    void *out, int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp,
                    int MBT, int N, hipStream_t st)
{
    using C = GemmCfg<L, WM, WN, NF, DEPTH, BDEPTH>;
    auto kern = gemm_i8_kernel<L, WM, WN, NF, DEPTH, BDEPTH, F8>;
    static PerDeviceFlag attr;
    if (!ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), C::LDS_BYTES)) return FQL_ERR_LAUNCH;
    (void)hipGetLastError();                                 // a stale error of another library must not read as ours
    // persistent: the 8-wave workgroups fill a CU alone; the small skinny-tile workgroups share it 4 / 8 ways
    const int cus = compute_units() * (C::NW >= 8 ? 1 : (C::NW == 4 ? 2 : 4));      // 2 waves per SIMD either way
    const int m_slots = (tpe == nullptr) ? (T + C::BM - 1) / C::BM : T / C::BM + E;
    const int groups = (tpe == nullptr) ? 1 : E;
    const long long m_even = (long long)groups * (((T + groups - 1) / groups + C::BM - 1) / C::BM);   // row blocks if evenly routed
    const int n_tiles = (N + C::BN - 1) / C::BN;            // the fewest column tiles that cover N ...
    int n_alt = (NF >= 2 && !F8) ? balanced_n_tiles(N, C::BN / 32, m_even, cus) : n_tiles;   // ... and the balanced alternative;
    if (n_alt == n_tiles) n_alt = 0;                         // the kernel picks between them from the real row-block count
    long long blocks = (long long)(n_alt > n_tiles ? n_alt : n_tiles) * m_slots;   // worst-case tile count (real count is on the device)
    if (blocks <= 0 || blocks > 0x7fffffffLL) return FQL_ERR_BAD_SHAPE;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, st, w.limbs, w.delta, w.rowsum,
                       packed, scales, zps, out, out_dtype | (w.row_weight != nullptr ? 8 : 0), tpe, offs, E, T, K, Kp, MBT, N, n_tiles, m_slots, w.scratch, w.bias, n_alt);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

template <int L, int NF, int KG, int DEPTH, int BDEPTH, int OCC>
int launch_rows32_cfg(const Workspace &w, const uint8_t *packed, const float *scales, const float *zps, void *out,
                      int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp, int MBT,
                      int N, hipStream_t st)
{
    using C = Rows32Cfg<L, NF, KG, DEPTH, BDEPTH, OCC>;
    auto kern = gemm_i8_rows32_kernel<L, NF, KG, DEPTH, BDEPTH, OCC>;
    static PerDeviceFlag attr;
    if (!ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), C::LDS_BYTES)) return FQL_ERR_LAUNCH;
    (void)hipGetLastError();                                 // a stale error of another library must not read as ours
    const int n_tiles = (N + C::BN - 1) / C::BN;
    const int m_slots = (tpe == nullptr) ? (T + C::BM - 1) / C::BM : T / C::BM + E;
    long long blocks = (long long)n_tiles * m_slots;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return FQL_ERR_BAD_SHAPE;
    const int cus = compute_units() * C::WG_PER_CU;          // persistent: WG_PER_CU 8-wave workgroups per CU
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, st, w.limbs, w.delta, w.rowsum,
                       packed, scales, zps, out, out_dtype | (w.row_weight != nullptr ? 8 : 0), tpe, offs, E, T, K, Kp, MBT, N, n_tiles, m_slots, w.scratch, w.bias);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

template <int L, int NF, int KG, int BDEPTH, int NWAVES = 8>
int launch_rows16_cfg(const Workspace &w, const uint8_t *packed, const float *scales, const float *zps, void *out,
                      int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp, int MBT,
                      int N, hipStream_t st)
{
    using C = Rows16Cfg<L, NF, KG, BDEPTH, NWAVES>;
    auto kern = gemm_i8_rows16_kernel<L, NF, KG, BDEPTH, NWAVES>;
    static PerDeviceFlag attr;
    if (!ensure_lds_attr(attr, reinterpret_cast<const void *>(kern), C::LDS_BYTES)) return FQL_ERR_LAUNCH;
    (void)hipGetLastError();                                 // a stale error of another library must not read as ours
    const int n_tiles = (N + C::BN - 1) / C::BN;
    const int m_slots = (tpe == nullptr) ? (T + C::BM - 1) / C::BM : T / C::BM + E;
    const int cus = compute_units() * (8 / C::NW);          // persistent: one 8-wave or two 4-wave workgroups per CU
    const int groups = (tpe == nullptr) ? 1 : E;
    const long long m_even = (long long)groups * (((T + groups - 1) / groups + C::BM - 1) / C::BM);   // row blocks if evenly routed
    int n_alt = balanced_n_tiles(N, C::BN / 16, m_even, cus, 16, 2, 1);     // uneven column tiles (fql_gemm_rows16.h)
    if (n_alt == n_tiles) n_alt = 0;
    long long blocks = (long long)(n_alt > n_tiles ? n_alt : n_tiles) * m_slots;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return FQL_ERR_BAD_SHAPE;
    if (blocks > cus) blocks = cus;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(C::THREADS), C::LDS_BYTES, st, w.limbs, w.delta, w.rowsum,
                       packed, scales, zps, out, out_dtype | (w.row_weight != nullptr ? 8 : 0), tpe, offs, E, T, K, Kp, MBT, N, n_tiles, m_slots, w.scratch, w.bias, n_alt);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

// One launch for pre-pass + GEMM (fql_gemm_w4.h, FUSED): tuning switch, the polls a workgroup spends on another one's rows
// before it quantises them itself, and the launch token (unique per launch: a flag word of an earlier launch, or of
// whatever the workspace held before, never equals it -- 2^-64 for arbitrary memory).
int g_fused = 0;
int g_fused_spin = 20000;
inline unsigned long long next_fused_token()
{
    static const unsigned long long salt = ((unsigned long long)std::random_device{}() << 32) | 0x100000000ull;
    static std::atomic<unsigned> counter{1};
    return salt ^ (unsigned long long)counter.fetch_add(1, std::memory_order_relaxed);
}
// The one-launch form keeps its tile list in the 16-entry LDS table: every workgroup must get by with one table.
inline bool w4_fusable(int nf, const int32_t *tpe, int E, int T, int N)
{
    const int BM = 128, BN = 32 * nf;
    const int cus = compute_units();
    const int m_slots = (tpe == nullptr) ? (T + BM - 1) / BM : T / BM + E;
    const int groups = (tpe == nullptr) ? 1 : E;
    const long long m_even = (long long)groups * (((T + groups - 1) / groups + BM - 1) / BM);
    const int n_tiles = (N + BN - 1) / BN;
    const int n_alt = balanced_n_tiles(N, BN / 32, m_even, cus);
    long long worst = (long long)(n_alt > n_tiles ? n_alt : n_tiles) * m_slots;
    if (worst <= 0) return false;
    const long long blocks = worst > cus ? cus : worst;
    return (worst + blocks - 1) / blocks <= 16;
}

// The one-wave-per-SIMD kernel: same tiles and column split as the wide kernel's 128 x 192 configuration.
int launch_w4_cfg(int L, int nf, int depth, const Workspace &w, const uint8_t *packed, const float *scales, const float *zps,
                  void *out, int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp, int MBT,
                  int N, hipStream_t st)
{
    if (Kp < 2 * FQL_KB) return FQL_ERR_BAD_SHAPE;           // its pipeline runs two weight stages ahead
    const int BM = 128, BN = fql_w4_bn(L, nf);
    const int cus = compute_units();
    const int m_slots = (tpe == nullptr) ? (T + BM - 1) / BM : T / BM + E;
    const int groups = (tpe == nullptr) ? 1 : E;
    const long long m_even = (long long)groups * (((T + groups - 1) / groups + BM - 1) / BM);
    const int n_tiles = (N + BN - 1) / BN;
    int n_alt = balanced_n_tiles(N, BN / 32, m_even, cus);
    if (n_alt == n_tiles) n_alt = 0;
    long long blocks = (long long)(n_alt > n_tiles ? n_alt : n_tiles) * m_slots;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return FQL_ERR_BAD_SHAPE;
    if (blocks > cus) blocks = cus;
    FqlW4Args a;
    a.limbs = w.limbs; a.delta = w.delta; a.rowsum = w.rowsum;
    a.packed = packed; a.scales = scales; a.zps = zps;
    a.out = out; a.out_kind = out_dtype | (w.row_weight != nullptr ? 8 : 0);
    a.tpe = tpe; a.offs = offs;
    a.E = E; a.T = T; a.K = K; a.Kp = Kp; a.MBT = MBT; a.N = N;
    a.n_tiles = n_tiles; a.m_slots = m_slots; a.n_alt = n_alt;
    a.scratch = w.scratch; a.bias = w.bias;
    a.blocks = blocks; a.stream = st;
    if (w.fuse_x != nullptr) {                               // one launch: the kernel quantises the rows itself (w4_fusable() said it may)
        a.fused = true;
        a.fz = FqlW4Fused{w.fuse_x, w.fuse_gather, w.fuse_n_src, w.row_weight, w.flags, next_fused_token(), g_fused_spin};
    }
    const int rc = fql_w4_launch(L, nf, depth, a);
    return rc == 0 ? FQL_OK : (rc == -2 ? FQL_ERR_BAD_SHAPE : FQL_ERR_LAUNCH);
}

template <int L>
int launch_gemm(int cfg, const Workspace &w, const uint8_t *packed, const float *scales, const float *zps,
                void *out, int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp, int MBT,
                int N, hipStream_t st)
{
    switch (cfg) {
#define W(i, l, nf, d)                                                                                            \
    case 300 + i:                                                                                                 \
        if (L != l) return FQL_ERR_BAD_SHAPE;                                                                      \
        return launch_w4_cfg(l, nf, d, w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
        FQL_W4_LIST(W)
#undef W
#define X(id, wm, wn, nf, d, bp)                                                                                  \
    case id:                                                                                                      \
        if constexpr (wide_cfg_built(L, id))                                                                      \
            return launch_gemm_cfg<L, wm, wn, nf, d, bp>(w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, \
                                                         Kp, MBT, N, st);                                         \
        else return FQL_ERR_BAD_SHAPE;
        FQL_CFG_LIST(X)
#undef X
#define R(i, nf, kg, d, bd, occ)                                                                                   \
    case 100 + i:                                                                                                 \
        return launch_rows32_cfg<L, nf, kg, d, bd, occ>(w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, \
                                                        Kp, MBT, N, st);
        FQL_ROWS32_LIST(R)
#undef R
#define S(i, nf, kg, bd)                                                                                           \
    case 200 + i:                                                                                                 \
        return launch_rows16_cfg<L, nf, kg, bd>(w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, \
                                                st);
        FQL_ROWS16_LIST(S)
#undef S
#define S4(i, nf, kg, bd)                                                                                          \
    case 220 + i:                                                                                                 \
        return launch_rows16_cfg<L, nf, kg, bd, 4>(w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, \
                                                   N, st);
        FQL_ROWS16_W4_LIST(S4)
#undef S4
    default: return FQL_ERR_BAD_SHAPE;
    }
}

// fp8-activation form of the wide kernel (gemm_i8_kernel<1, ..., F8 = true>): the configurations it is built for.
#define FQL_F8_CFG_LIST(Y)                                                                                         \
    Y(1, 4, 2, 2, 4, 1)        /* 128 x 128 */ \
    Y(5, 1, 8, 1, 4, 1)        /*  32 x 256 */ \
    Y(6, 4, 1, 2, 8, 4)        /* 128 x  64, 4 waves */ \
    Y(7, 2, 2, 1, 8, 4)        /*  64 x  64, 4 waves */ \
    Y(8, 1, 2, 1, 8, 4)        /*  32 x  64, 2 waves */ \
    Y(11, 2, 4, 3, 8, 2)       /*  64 x 384, full-stage activation ring, 2 weight stages */ \
    Y(12, 4, 2, 3, 8, 2)       /* 128 x 192, full-stage activation ring, 2 weight stages */
inline bool valid_cfg_f8(int cfg) { return cfg == 1 || cfg == 5 || cfg == 6 || cfg == 7 || cfg == 8 || cfg == 11 || cfg == 12; }

int launch_gemm_f8(int cfg, const Workspace &w, const uint8_t *packed, const float *scales, const float *zps, void *out,
                   int out_dtype, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int Kp, int MBT, int N,
                   hipStream_t st)
{
    switch (cfg) {
#define Y(id, wm, wn, nf, d, bp)                                                                                  \
    case id:                                                                                                      \
        return launch_gemm_cfg<1, wm, wn, nf, d, bp, true>(w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, \
                                                           Kp, MBT, N, st);
        FQL_F8_CFG_LIST(Y)
#undef Y
    default: return FQL_ERR_BAD_SHAPE;
    }
}

// Tile choice of the fp8 form: weight-streaming bound at every shape it is meant for (one MFMA pass).
inline int choose_cfg_f8(int E, int T, int N, bool grouped)
{
    const int groups = grouped ? (E > 0 ? E : 1) : 1;
    const int m = (T + groups - 1) / groups;
    const long long wide_tiles = (long long)groups * ((m + 127) / 128) * ((N + 255) / 256);
    if (wide_tiles < 128 && m <= 128) return m <= 32 ? 8 : (m <= 64 ? 7 : 6);
    if (m <= 32) return 5;
    if (m <= 64) return 11;
    return 12;
}

// Heuristic tile choice for the product path (MI355X: 256 CUs, one workgroup per CU).
//   rows per group decide the tile height (32 / 64 / 128-row tiles: a short group must not pay for
//   128 rows of MFMA work); the tile width is the one that needs the least "rounds x width" of the
//   256 CUs -- e.g. 8 experts x 128 rows x N = 11008: 128 x 192 tiles give 464 tiles = 2 rounds x 192,
//   128 x 128 give 688 = 3 rounds x 128 (equal), 128 x 256 give 344 = 2 rounds x 256 (worse).
inline int choose_cfg(int L, int E, int T, int K, int N, bool grouped)
{
    (void)K;
    const int groups = grouped ? (E > 0 ? E : 1) : 1;
    const int m = (T + groups - 1) / groups;                 // rows per group if evenly routed
    // few rows in total: the op is HBM-bound, what matters is enough tiles to have every CU streaming
    // weights (N / 64 tiles per row block instead of N / 256)
    const long long wide_tiles = (long long)groups * ((m + 127) / 128) * ((N + 255) / 256);
    if (wide_tiles < 128 && m <= 128) {
        if (m <= 16) return 207;                             //  16 x 48 decode tiles, K split 8 ways (fql_gemm_rows16.h)
        if (m <= 32) return 8;                               //  32 x 64, 2 waves
        if (m <= 64) return 7;                               //  64 x 64, 4 waves
        // 65..128 rows: 128 x 64, 4 waves, 8 weight stages in flight; at 3 limbs the 128 x 128 tile of the wide kernel is
        // 8-10 % faster (96 rows 38.1 vs 41.3 us, 128 rows 40.3 vs 44.5 us, 4096 -> 11008, profiles/r03_small_groups.txt)
        return L == 3 ? 1 : 13;
    }
    if (m <= 16) return 202;                                 //  16 x 128 decode tiles: every load one stage ahead
    if (m <= 32) return 103;                                 //  32 x 128, K split 4 ways inside the workgroup
    if (m <= 64) {                                           //  64 x 384 (1 / 2 limbs: full-stage activation ring) ...
        // ... 3 limbs: the one-wave-per-SIMD kernel's 33..64-row tile class (two row blocks x two fragment halves per
        // workgroup): 79.7 vs 89.3 us at 8 x 64 rows (profiles/r03_small_groups.txt); at 32 rows the 32-row kernel stays (67 vs 70)
        if (L == 3 && g_use_w4 && padded_k(K) >= 2 * FQL_KB) return 301;
        return (L <= 2) ? 11 : 9;
    }
    const int mt = groups * ((m + 127) / 128);
    struct Cand { int cfg, bn; };
    // 3 limbs: 128 x 192 with a 2-step A ring (register budget) / 128 x 128; 2 limbs: 4-step ring, + 128 x 256
    const Cand c3[2] = {{0, 192}, {1, 128}};
    const Cand c2[3] = {{3, 192}, {1, 128}, {2, 256}};
    const Cand c1[3] = {{12, 192}, {1, 128}, {2, 256}};      // 1 limb: registers allow the full-stage activation ring
    const Cand *cands = (L == 1) ? c1 : (L == 2) ? c2 : c3;
    const int nc = (L <= 2) ? 3 : 2;
    int best = cands[0].cfg;
    long long best_cost = -1;
    for (int i = 0; i < nc; ++i) {
        const long long tiles = (long long)mt * ((N + cands[i].bn - 1) / cands[i].bn);
        const long long rounds = (tiles + 255) / 256;
        const long long cost = rounds * (cands[i].bn + 24);  // +24: per-tile prologue / epilogue
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cands[i].cfg; }
    }
    // the same 128 x 192 tiles at one wave per SIMD (measured: 124.7 vs 134.4 us at configs[2], 182.6 vs 204.4 us under
    // skewed routing, profiles/r03_w4_vs_wide.txt); its pipeline runs two 256-k weight stages ahead
    if (best == 0 && g_use_w4 && padded_k(K) >= 2 * FQL_KB) best = 301;
    return best;
}

int run_mfma(int L, const void *x, int in_dtype, const int32_t *gather, int n_src, const uint8_t *packed,
             const float *scales, const float *zps, void *out, int out_dtype, const int32_t *tpe, const int32_t *offs,
             int E, int T, int K, int N, void *workspace, size_t workspace_bytes, hipStream_t st, bool gated = false,
             bool f8 = false, const float *bias = nullptr, const float *row_weight = nullptr)
{
    const int Kp = padded_k(K);
    const int MBT = row_blocks(T, E);
    if (workspace == nullptr || !aligned16(workspace)) return FQL_ERR_WORKSPACE;
    Workspace w = carve(workspace, L, T, E, Kp, has_residual(L, f8));
    if (workspace_bytes < w.bytes) return FQL_ERR_WORKSPACE;
    w.bias = bias;
    w.row_weight = row_weight;
    void *zero_out = (tpe != nullptr) ? out : nullptr;
    int rc;
    if (f8) {                                                // float rows -> e4m3 with a per-row scale, one fp8 MFMA pass
        if (gated) return FQL_ERR_BAD_PRECISION;
        rc = launch_act_quant<1>(x, in_dtype, gather, n_src, w, T, K, Kp, MBT, zero_out, out_dtype, N, tpe, offs, E, st, false, true);
        if (rc != FQL_OK) return rc;
        return launch_gemm_f8(choose_cfg_f8(E, T, N, tpe != nullptr), w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
    }
    const int cfg = choose_cfg(L, E, T, K, N, tpe != nullptr);
    if (L == 1) {
        rc = launch_act_quant<1>(x, in_dtype, gather, n_src, w, T, K, Kp, MBT, zero_out, out_dtype, N, tpe, offs, E, st, gated);
        if (rc != FQL_OK) return rc;
        return launch_gemm<1>(cfg, w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
    }
    if (L == 2) {
        rc = launch_act_quant<2>(x, in_dtype, gather, n_src, w, T, K, Kp, MBT, zero_out, out_dtype, N, tpe, offs, E, st, gated);
        if (rc != FQL_OK) return rc;
        return launch_gemm<2>(cfg, w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
    }
    if (g_fused && cfg == 301 && in_dtype == FQL_DTYPE_F32 && !gated && w.flags != nullptr && (K % 16 == 0) && aligned16(x) &&
        w4_fusable(6, tpe, E, T, N)) {
        w.fuse_x = x; w.fuse_gather = gather; w.fuse_n_src = n_src;      // no pre-pass launch: the GEMM kernel's first phase
        return launch_gemm<3>(cfg, w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
    }
    rc = launch_act_quant<3>(x, in_dtype, gather, n_src, w, T, K, Kp, MBT, zero_out, out_dtype, N, tpe, offs, E, st, gated);
    if (rc != FQL_OK) return rc;
    return launch_gemm<3>(cfg, w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
}

int run_generic(const float *x, const uint8_t *packed, const float *scales, const float *zps, float *out,
                const int32_t *tpe, const int32_t *offs, int E, int T, int K, int N, hipStream_t st,
                const float *bias = nullptr)
{
    if (tpe != nullptr) {
        hipLaunchKernelGGL(zero_uncovered_rows_kernel, dim3(T), dim3(256), 0, st, out, tpe, offs, E, T, N);
        if (hipGetLastError() != hipSuccess) return FQL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL((fused_rows_kernel<4>), dim3((N + 3) / 4, E), dim3(256), 0, st, x, packed, scales, zps,
                       out, tpe, offs, T, K, N, bias);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

size_t gemv_lds_bytes(int B, int K) { return ((size_t)B * (K >> 5) * GEMV_SEG + (size_t)B * 4) * sizeof(float); }

template <int B, bool GROUPED = false>
int launch_gemv(const float *x, const uint8_t *packed, const float *scales, const float *zps, float *out,
                int K, int N, hipStream_t st, const float *bias, int group = 0)
{
    const int groups = (N + GEMV_ROWS - 1) / GEMV_ROWS;
    int blocks = (groups + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const size_t lds = gemv_lds_bytes(B, K);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&gemv_kernel<B, GROUPED>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return FQL_ERR_LAUNCH;
    }
    (void)hipGetLastError();
    hipLaunchKernelGGL((gemv_kernel<B, GROUPED>), dim3(blocks), dim3(256), lds, st, x, packed, scales, zps, out, K, N, bias, group);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

}  // namespace

static bool valid_dtype(int dt) { return dt == FQL_DTYPE_F32 || dt == FQL_DTYPE_F16 || dt == FQL_DTYPE_BF16; }

extern "C" {

int fql_version(void) { return FQL_VERSION; }

const char *fql_error_string(int code)
{
    switch (code) {
    case FQL_OK: return "ok";
    case FQL_ERR_NULL_POINTER: return "a required pointer is NULL";
    case FQL_ERR_BAD_SHAPE: return "bad shape: dimensions must be positive and fit the addressable range";
    case FQL_ERR_ODD_K: return "input_dim (K) must be even: two 4-bit weights per packed byte";
    case FQL_ERR_WORKSPACE: return "workspace is NULL, not 16-byte aligned, or smaller than *_workspace_bytes()";
    case FQL_ERR_LAUNCH: return "kernel launch failed (hipGetLastError)";
    case FQL_ERR_BAD_PRECISION: return "precision must be FQL_PRECISION_DEFAULT, _INT8 (1), _FAST (2), _EXACT (3) or _FP8 (8), and is supported by this entry point";
    case FQL_ERR_DTYPE: return "element type not supported on this path (16-bit input / output exists on the MFMA path only)";
    case FQL_ERR_ALIGNMENT: return "tensor base pointer not aligned as documented";
    default: return "unknown error code";
    }
}

int fql_act_padded_k(int K) { return K > 0 ? padded_k(K) : 0; }

size_t fql_linear_workspace_bytes(int B, int K, int N, int precision)
{
    (void)N;
    const int L = limbs_of(precision);
    if (L < 0 || (B <= g_gemv_max_rows && !is_f8(precision)) || B <= 0 || K <= 0 || (K % 32) != 0) return 0;   // (GEMV shapes use no workspace; fql_linear_fwd_f8 takes any B on the MFMA path)
    Workspace w = carve(nullptr, L, B, 1, padded_k(K), has_residual(L, is_f8(precision)));
    return w.bytes;
}

size_t fql_moe_workspace_bytes(int E, int T, int K, int N, int precision)
{
    (void)N;
    const int L = limbs_of(precision);
    if (L < 0 || T <= 0 || E <= 0 || K <= 0 || (K % 32) != 0) return 0;
    Workspace w = carve(nullptr, L, T, E, padded_k(K), has_residual(L, is_f8(precision)));
    return w.bytes;
}

static int linear_f32_core(const float *x, const uint8_t *packed, const float *scales, const float *zps, const float *bias,
                           float *out, int B, int K, int N, int precision, void *workspace,
                           size_t workspace_bytes, void *stream)
{
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (B < 0 || K < 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (B == 0 || N == 0) return FQL_OK;
    if (!x || !packed || !scales || !zps || !out) return FQL_ERR_NULL_POINTER;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (K == 0) {                                     // empty contraction: out = 0 (+ bias: the caller adds it; not a path worth a kernel)
        if (bias != nullptr) return FQL_ERR_BAD_SHAPE;
        return hipMemsetAsync(out, 0, (size_t)B * N * sizeof(float), st) == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
    }
    const bool mfma_small = B > g_gemv_max_rows && mfma_eligible(L, B, 1, K, N, packed) && workspace != nullptr;
    if (is_f8(precision) && (B <= g_gemv_max_rows || !mfma_eligible(L, B, 1, K, N, packed)))
        return FQL_ERR_BAD_PRECISION;                 // fp8 exists on the MFMA path only: no silent float32 fallback
    if (is_f8(precision) && !mfma_small) return FQL_ERR_WORKSPACE;
    if (B <= 4 && !mfma_small) {
        if ((K % 32 == 0) && aligned16(packed) && aligned16(x) && gemv_lds_bytes(B, K) <= 150 * 1024) {
            switch (B) {
            case 1: return launch_gemv<1>(x, packed, scales, zps, out, K, N, st, bias);
            case 2: return launch_gemv<2>(x, packed, scales, zps, out, K, N, st, bias);
            case 3: return launch_gemv<3>(x, packed, scales, zps, out, K, N, st, bias);
            default: return launch_gemv<4>(x, packed, scales, zps, out, K, N, st, bias);
            }
        }
        return run_generic(x, packed, scales, zps, out, nullptr, nullptr, 1, B, K, N, st, bias);
    }
    if (mfma_eligible(L, B, 1, K, N, packed))
        return run_mfma(L, x, FQL_DTYPE_F32, nullptr, 0, packed, scales, zps, out, FQL_DTYPE_F32, nullptr, nullptr, 1, B, K, N,
                        workspace, workspace_bytes, st, false, is_f8(precision), bias);
    return run_generic(x, packed, scales, zps, out, nullptr, nullptr, 1, B, K, N, st, bias);
}

int fql_linear_fwd_f32(const float *x, const uint8_t *packed, const float *scales, const float *zps,
                       float *out, int B, int K, int N, int precision, void *workspace,
                       size_t workspace_bytes, void *stream)
{
    return linear_f32_core(x, packed, scales, zps, nullptr, out, B, K, N, precision, workspace, workspace_bytes, stream);
}

int fql_linear_bias_fwd_f32(const float *x, const uint8_t *packed, const float *scales, const float *zps,
                            const float *bias, float *out, int B, int K, int N, int precision, void *workspace,
                            size_t workspace_bytes, void *stream)
{
    return linear_f32_core(x, packed, scales, zps, bias, out, B, K, N, precision, workspace, workspace_bytes, stream);
}


static int moe_entry(const uint8_t *packed, const float *scales, const float *zps, const float *inputs,
                     const int32_t *row_index, int n_src, const int32_t *tokens_per_expert,
                     const int32_t *input_offsets, float *out, int E, int T, int K, int N, int precision,
                     void *workspace, size_t workspace_bytes, void *stream, const float *row_weight = nullptr)
{
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (E < 0 || T < 0 || K < 0 || N < 0 || (row_index != nullptr && n_src <= 0)) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (T == 0 || N == 0) return FQL_OK;
    if (!out) return FQL_ERR_NULL_POINTER;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (E == 0 || K == 0) {                           // nothing contributes: all rows zero
        return hipMemsetAsync(out, 0, (size_t)T * N * sizeof(float), st) == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
    }
    if (!packed || !scales || !zps || !inputs || !tokens_per_expert || !input_offsets) return FQL_ERR_NULL_POINTER;
    if (E > 65535) return FQL_ERR_BAD_SHAPE;
    if (mfma_eligible(L, T, E, K, N, packed))
        return run_mfma(L, inputs, FQL_DTYPE_F32, row_index, n_src, packed, scales, zps, out, FQL_DTYPE_F32,
                        tokens_per_expert, input_offsets, E, T, K, N, workspace, workspace_bytes, st, false, is_f8(precision),
                        nullptr, row_weight);
    if (row_index != nullptr || is_f8(precision) || row_weight != nullptr) return FQL_ERR_ALIGNMENT;     // the fused gather / row weights / fp8 exist on the MFMA path only
    return run_generic(inputs, packed, scales, zps, out, tokens_per_expert, input_offsets, E, T, K, N, st);
}

int fql_moe_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *inputs,
                    const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                    int K, int N, int precision, void *workspace, size_t workspace_bytes, void *stream)
{
    return moe_entry(packed, scales, zps, inputs, nullptr, 0, tokens_per_expert, input_offsets, out, E, T, K, N,
                     precision, workspace, workspace_bytes, stream);
}

int fql_moe_gather_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *tokens,
                           const int32_t *row_index, int n_tokens, const int32_t *tokens_per_expert,
                           const int32_t *input_offsets, float *out, int E, int T, int K, int N, int precision,
                           void *workspace, size_t workspace_bytes, void *stream)
{
    if (!row_index) return FQL_ERR_NULL_POINTER;
    return moe_entry(packed, scales, zps, tokens, row_index, n_tokens, tokens_per_expert, input_offsets, out, E, T, K,
                     N, precision, workspace, workspace_bytes, stream);
}


int fql_moe_gather_scaled_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *tokens,
                                  const int32_t *row_index, int n_tokens, const float *row_weight,
                                  const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                                  int K, int N, int precision, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!row_index || !row_weight) return FQL_ERR_NULL_POINTER;
    if (is_f8(precision)) return FQL_ERR_BAD_PRECISION;
    return moe_entry(packed, scales, zps, tokens, row_index, n_tokens, tokens_per_expert, input_offsets, out, E, T, K,
                     N, precision, workspace, workspace_bytes, stream, row_weight);
}

int fql_native_dtype_supported(int rows, int E, int K, int N, int precision, const void *packed, int grouped)
{
    const int L = limbs_of(precision);
    if (L < 0 || rows <= 0 || E <= 0) return 0;
    if (!grouped && rows <= g_gemv_max_rows) return 0;       // the GEMV path is float32 only
    return mfma_eligible(L, rows, E, K, N, static_cast<const uint8_t *>(packed)) ? 1 : 0;
}

int fql_linear_fwd(const void *x, int in_dtype, const uint8_t *packed, const float *scales, const float *zps, void *out,
                   int out_dtype, int B, int K, int N, int precision, void *workspace, size_t workspace_bytes,
                   void *stream)
{
    if (in_dtype == FQL_DTYPE_F32 && out_dtype == FQL_DTYPE_F32)
        return fql_linear_fwd_f32(static_cast<const float *>(x), packed, scales, zps, static_cast<float *>(out), B, K, N,
                                  precision, workspace, workspace_bytes, stream);
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (!valid_dtype(in_dtype) || !valid_dtype(out_dtype)) return FQL_ERR_DTYPE;
    if (B < 0 || K < 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (B == 0 || N == 0) return FQL_OK;
    if (!x || !packed || !scales || !zps || !out) return FQL_ERR_NULL_POINTER;
    if (B <= g_gemv_max_rows || !mfma_eligible(L, B, 1, K, N, packed)) return FQL_ERR_DTYPE;   // 16-bit I/O exists on the MFMA path only
    return run_mfma(L, x, in_dtype, nullptr, 0, packed, scales, zps, out, out_dtype, nullptr, nullptr, 1, B, K, N,
                    workspace, workspace_bytes, static_cast<hipStream_t>(stream), false, is_f8(precision));
}

int fql_moe_fwd(const uint8_t *packed, const float *scales, const float *zps, const void *inputs, int in_dtype,
                const int32_t *tokens_per_expert, const int32_t *input_offsets, void *out, int out_dtype, int E, int T,
                int K, int N, int precision, void *workspace, size_t workspace_bytes, void *stream)
{
    if (in_dtype == FQL_DTYPE_F32 && out_dtype == FQL_DTYPE_F32)
        return fql_moe_fwd_f32(packed, scales, zps, static_cast<const float *>(inputs), tokens_per_expert, input_offsets,
                               static_cast<float *>(out), E, T, K, N, precision, workspace, workspace_bytes, stream);
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (!valid_dtype(in_dtype) || !valid_dtype(out_dtype)) return FQL_ERR_DTYPE;
    if (E <= 0 || T < 0 || K <= 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (T == 0 || N == 0) return FQL_OK;
    if (!packed || !scales || !zps || !inputs || !tokens_per_expert || !input_offsets || !out) return FQL_ERR_NULL_POINTER;
    if (E > 65535) return FQL_ERR_BAD_SHAPE;
    if (!mfma_eligible(L, T, E, K, N, packed)) return FQL_ERR_DTYPE;
    return run_mfma(L, inputs, in_dtype, nullptr, 0, packed, scales, zps, out, out_dtype, tokens_per_expert,
                    input_offsets, E, T, K, N, workspace, workspace_bytes, static_cast<hipStream_t>(stream), false, is_f8(precision));
}

// ---- rows that are already OCP e4m3 (BASELINE.json configs[4]): re-layout pre-pass + one fp8 MFMA pass
static int f8_entry(const uint8_t *packed, const float *scales, const float *zps, const uint8_t *x8, const float *act_scales,
                    const int32_t *tpe, const int32_t *offs, void *out, int out_dtype, int E, int T, int K, int N,
                    void *workspace, size_t workspace_bytes, void *stream)
{
    if (!valid_dtype(out_dtype)) return FQL_ERR_DTYPE;
    if (E <= 0 || T < 0 || K <= 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (T == 0 || N == 0) return FQL_OK;
    if (!packed || !scales || !zps || !x8 || !out) return FQL_ERR_NULL_POINTER;
    if ((tpe == nullptr) != (offs == nullptr)) return FQL_ERR_NULL_POINTER;
    if (tpe == nullptr && E != 1) return FQL_ERR_BAD_SHAPE;
    if (E > 65535) return FQL_ERR_BAD_SHAPE;
    if (!mfma_eligible(1, T, E, K, N, packed)) return FQL_ERR_ALIGNMENT;      // K % 32 == 0, 16-byte aligned weights
    const int Kp = padded_k(K), MBT = row_blocks(T, E);
    if (workspace == nullptr || !aligned16(workspace)) return FQL_ERR_WORKSPACE;
    const Workspace w = carve(workspace, 1, T, E, Kp, false);
    if (workspace_bytes < w.bytes) return FQL_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int mblocks = (tpe == nullptr) ? (T + FQL_MB - 1) / FQL_MB : (T + FQL_MB * E) / FQL_MB;
    const int rblocks = mblocks * (FQL_MB / ACT_ROWS);
    const int zblocks = (tpe != nullptr) ? (T + 255) / 256 : 0;
    const int vec = ((K % 16) == 0 && (reinterpret_cast<uintptr_t>(x8) % 16) == 0) ? 1 : 0;
    (void)hipGetLastError();
    hipLaunchKernelGGL(act_f8_relayout_kernel, dim3(rblocks + zblocks), dim3(256), 0, st, x8, act_scales, (const int32_t *)nullptr, 0,
                       w.delta, w.rowsum, w.limbs, T, K, Kp / FQL_KB, MBT, rblocks, out, dtype_bytes(out_dtype), N, tpe, offs, E, vec);
    if (hipGetLastError() != hipSuccess) return FQL_ERR_LAUNCH;
    return launch_gemm_f8(choose_cfg_f8(E, T, N, tpe != nullptr), w, packed, scales, zps, out, out_dtype, tpe, offs, E, T, K, Kp, MBT, N, st);
}

int fql_moe_fwd_f8(const uint8_t *packed, const float *scales, const float *zps, const uint8_t *inputs_e4m3,
                   const float *act_scales, const int32_t *tokens_per_expert, const int32_t *input_offsets, void *out,
                   int out_dtype, int E, int T, int K, int N, void *workspace, size_t workspace_bytes, void *stream)
{
    if (!tokens_per_expert || !input_offsets) return FQL_ERR_NULL_POINTER;
    return f8_entry(packed, scales, zps, inputs_e4m3, act_scales, tokens_per_expert, input_offsets, out, out_dtype, E, T, K, N,
                    workspace, workspace_bytes, stream);
}

int fql_linear_fwd_f8(const uint8_t *x_e4m3, const float *act_scales, const uint8_t *packed, const float *scales,
                      const float *zps, void *out, int out_dtype, int B, int K, int N, void *workspace,
                      size_t workspace_bytes, void *stream)
{
    return f8_entry(packed, scales, zps, x_e4m3, act_scales, nullptr, nullptr, out, out_dtype, 1, B, K, N, workspace,
                    workspace_bytes, stream);
}

// ---- per-group scales along K (functional path: csrc/fql_generic.h)
static int group_entry(const float *x, const uint8_t *packed, const float *scales, const float *zps, const float *bias,
                       float *out, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int N, int group,
                       void *stream)
{
    if (E <= 0 || T < 0 || K < 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (group <= 0 || (group & 1) || K % group != 0) return FQL_ERR_BAD_SHAPE;     // even groups that tile K
    if (T == 0 || N == 0) return FQL_OK;
    if (!x || !packed || !scales || !zps || !out) return FQL_ERR_NULL_POINTER;
    if ((tpe == nullptr) != (offs == nullptr)) return FQL_ERR_NULL_POINTER;
    if (tpe == nullptr && E != 1) return FQL_ERR_BAD_SHAPE;
    if (E > 65535) return FQL_ERR_BAD_SHAPE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (K == 0) return hipMemsetAsync(out, 0, (size_t)T * N * sizeof(float), st) == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
    (void)hipGetLastError();
    if (tpe != nullptr) {
        hipLaunchKernelGGL(zero_uncovered_rows_kernel, dim3(T), dim3(256), 0, st, out, tpe, offs, E, T, N);
        if (hipGetLastError() != hipSuccess) return FQL_ERR_LAUNCH;
    }
    // batches: the float32 matrix-core kernel (fql_group.h); a few rows per group: one wave per output row (fql_generic.h)
    const int groups = tpe == nullptr ? 1 : E;
    const bool batch = g_group_mfma && (T + groups - 1) / groups >= 4 && K % 64 == 0 && group % 32 == 0 &&
                       (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (reinterpret_cast<uintptr_t>(packed) % 16 == 0) &&
                       (T + 63) / 64 <= 65535;
    const int per = (T + groups - 1) / groups;
    if (tpe == nullptr && T <= 3 && K % 32 == 0 && group % 32 == 0 && aligned16(packed) && aligned16(x) &&
        gemv_lds_bytes(T, K) <= 150 * 1024) {                     // a few rows of one matrix: the GEMV kernel, constants per group
        switch (T) {
        case 1: return launch_gemv<1, true>(x, packed, scales, zps, out, K, N, st, bias, group);
        case 2: return launch_gemv<2, true>(x, packed, scales, zps, out, K, N, st, bias, group);
        default: return launch_gemv<3, true>(x, packed, scales, zps, out, K, N, st, bias, group);
        }
    }
    if (batch && per <= 128 && K % 256 == 0 && (T + 31) / 32 <= 65535)   // few rows per group: 32 x 32 blocks, K split over the waves
        hipLaunchKernelGGL((group_mfma_kernel<true, 1>), dim3((N + 31) / 32, (T + 31) / 32, E), dim3(256), 0, st, x, packed, scales,
                           zps, out, tpe, offs, T, K, N, group, bias);
    else if (batch && (long long)((N + 127) / 128) * ((per + 63) / 64) * groups >= 2LL * compute_units())
        hipLaunchKernelGGL((group_mfma_kernel<false, 2>), dim3((N + 127) / 128, (T + 63) / 64, E), dim3(256), 0, st, x, packed,
                           scales, zps, out, tpe, offs, T, K, N, group, bias);
    else if (batch)
        hipLaunchKernelGGL((group_mfma_kernel<false, 1>), dim3((N + 63) / 64, (T + 63) / 64, E), dim3(256), 0, st, x, packed,
                           scales, zps, out, tpe, offs, T, K, N, group, bias);
    else
        hipLaunchKernelGGL((fused_rows_group_kernel<4>), dim3((N + 3) / 4, E), dim3(256), 0, st, x, packed, scales, zps, out,
                           tpe, offs, T, K, N, group, bias);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

// ---- per-group scales on the INTEGER matrix cores (csrc/fql_group_i8.h): needs the activation workspace of the per-row
//      path plus a [E][G][N] transpose of the scales and zero points
static bool group_i8_eligible(int L, int E, int T, int K, int N, int group, const void *x, const void *packed, bool grouped)
{
    const int groups = grouped ? E : 1;
    return g_group_i8 && L >= 1 && L <= 3 && K % FQL_KB == 0 && group % 64 == 0 && K % group == 0 && (T + groups - 1) / groups >= (grouped ? g_group_i8_min_rows_grouped : g_group_i8_min_rows) &&
           aligned16(packed) && (reinterpret_cast<uintptr_t>(x) % 4 == 0) && N >= 4 && (T + FQL_MB - 1) / FQL_MB + 1 <= 65535 && E <= 65535;
}

size_t fql_group_workspace_bytes(int E, int T, int K, int N, int group_size, int precision)
{
    const int L = limbs_of(precision);
    if (L < 1 || is_f8(precision) || E <= 0 || T <= 0 || K <= 0 || N <= 0 || group_size <= 0 || K % group_size != 0 || (K % 32) != 0) return 0;
    const Workspace w = carve(nullptr, L, T, E, padded_k(K), has_residual(L, false));
    return round16(w.bytes) + 2 * round16((size_t)E * N * (K / group_size) * sizeof(float));
}

static int group_ws_entry(const float *x, const uint8_t *packed, const float *scales, const float *zps, const float *bias,
                          float *out, const int32_t *tpe, const int32_t *offs, int E, int T, int K, int N, int group,
                          int precision, void *workspace, size_t workspace_bytes, void *stream)
{
    const int L = limbs_of(precision);
    if (L < 0 || is_f8(precision)) return FQL_ERR_BAD_PRECISION;
    if (E <= 0 || T < 0 || K < 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (group <= 0 || K <= 0 || K % group != 0) return (K == 0) ? group_entry(x, packed, scales, zps, bias, out, tpe, offs, E, T, K, N, group > 0 ? group : 2, stream) : FQL_ERR_BAD_SHAPE;
    const bool ok = workspace != nullptr && aligned16(workspace) && x && packed && scales && zps && out && T > 0 && N > 0 &&
                    ((tpe == nullptr) == (offs == nullptr)) && (tpe != nullptr || E == 1) &&
                    group_i8_eligible(L, E, T, K, N, group, x, packed, tpe != nullptr) &&
                    workspace_bytes >= fql_group_workspace_bytes(E, T, K, N, group, precision);
    if (!ok) return group_entry(x, packed, scales, zps, bias, out, tpe, offs, E, T, K, N, group, stream);   // float32 paths (and their checks)
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Kp = padded_k(K), MBT = row_blocks(T, E), G = K / group;
    Workspace w = carve(workspace, L, T, E, Kp, has_residual(L, false));
    float *st_t = reinterpret_cast<float *>(static_cast<char *>(workspace) + round16(w.bytes));
    float *zt_t = reinterpret_cast<float *>(reinterpret_cast<char *>(st_t) + round16((size_t)E * N * G * sizeof(float)));
    void *zero_out = (tpe != nullptr) ? out : nullptr;
    int rc;
    if (L == 1) rc = launch_act_quant<1>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, zero_out, FQL_DTYPE_F32, N, tpe, offs, E, st);
    else if (L == 2) rc = launch_act_quant<2>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, zero_out, FQL_DTYPE_F32, N, tpe, offs, E, st);
    else rc = launch_act_quant<3>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, zero_out, FQL_DTYPE_F32, N, tpe, offs, E, st);
    if (rc != FQL_OK) return rc;
    (void)hipGetLastError();
    const size_t total = (size_t)E * N * G;
    hipLaunchKernelGGL(transpose_ng_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, scales, zps, st_t, zt_t, N, G, total);
    if (hipGetLastError() != hipSuccess) return FQL_ERR_LAUNCH;
    const dim3 grid((N + 127) / 128, (T + FQL_MB - 1) / FQL_MB, E);
    if (L == 1) hipLaunchKernelGGL(group_i8_kernel<1>, grid, dim3(256), 0, st, w.limbs, w.delta, packed, st_t, zt_t, out, tpe, offs, E, T, K, MBT, N, group, bias, has_residual(L, false) ? 1 : 0);
    else if (L == 2) hipLaunchKernelGGL(group_i8_kernel<2>, grid, dim3(256), 0, st, w.limbs, w.delta, packed, st_t, zt_t, out, tpe, offs, E, T, K, MBT, N, group, bias, has_residual(L, false) ? 1 : 0);
    else hipLaunchKernelGGL(group_i8_kernel<3>, grid, dim3(256), 0, st, w.limbs, w.delta, packed, st_t, zt_t, out, tpe, offs, E, T, K, MBT, N, group, bias, has_residual(L, false) ? 1 : 0);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_linear_group_ws_fwd_f32(const float *x, const uint8_t *packed, const float *scales, const float *zps,
                                const float *bias, float *out, int B, int K, int N, int group_size, int precision,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    return group_ws_entry(x, packed, scales, zps, bias, out, nullptr, nullptr, 1, B, K, N, group_size, precision, workspace,
                          workspace_bytes, stream);
}

int fql_moe_group_ws_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *inputs,
                             const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                             int K, int N, int group_size, int precision, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    if (!tokens_per_expert || !input_offsets) return FQL_ERR_NULL_POINTER;
    return group_ws_entry(inputs, packed, scales, zps, nullptr, out, tokens_per_expert, input_offsets, E, T, K, N, group_size,
                          precision, workspace, workspace_bytes, stream);
}

int fql_linear_group_fwd_f32(const float *x, const uint8_t *packed, const float *scales, const float *zps,
                             const float *bias, float *out, int B, int K, int N, int group_size, void *stream)
{
    return group_entry(x, packed, scales, zps, bias, out, nullptr, nullptr, 1, B, K, N, group_size, stream);
}

int fql_moe_group_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *inputs,
                          const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                          int K, int N, int group_size, void *stream)
{
    if (!tokens_per_expert || !input_offsets) return FQL_ERR_NULL_POINTER;
    return group_entry(inputs, packed, scales, zps, nullptr, out, tokens_per_expert, input_offsets, E, T, K, N, group_size,
                       stream);
}

int fql_moe_gated_fwd_f32(const uint8_t *packed, const float *scales, const float *zps, const float *gate_up,
                          const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                          int K, int N, int precision, void *workspace, size_t workspace_bytes, void *stream)
{
    const int L = limbs_of(precision);
    if (L < 0 || is_f8(precision)) return FQL_ERR_BAD_PRECISION;
    if (E <= 0 || T < 0 || K <= 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (T == 0 || N == 0) return FQL_OK;
    if (!packed || !scales || !zps || !gate_up || !out) return FQL_ERR_NULL_POINTER;
    if ((tokens_per_expert == nullptr) != (input_offsets == nullptr)) return FQL_ERR_NULL_POINTER;
    if (tokens_per_expert == nullptr && E != 1) return FQL_ERR_BAD_SHAPE;
    if (E > 65535) return FQL_ERR_BAD_SHAPE;
    if (!mfma_eligible(L, T, E, K, N, packed)) return FQL_ERR_ALIGNMENT;   // the fused activation exists on the MFMA path only
    return run_mfma(L, gate_up, FQL_DTYPE_F32, nullptr, 0, packed, scales, zps, out, FQL_DTYPE_F32, tokens_per_expert,
                    input_offsets, E, T, K, N, workspace, workspace_bytes, static_cast<hipStream_t>(stream), true);
}

int fql_route_plan_i32(const int32_t *expert_of_slot, int n_slots, int top_k, int E, int32_t *counts,
                       int32_t *offsets, int32_t *token_of_sorted, int32_t *pos_of_slot, void *stream)
{
    if (n_slots < 0 || top_k <= 0 || E <= 0 || E > ROUTE_MAX_EXPERTS) return FQL_ERR_BAD_SHAPE;
    if (!counts || !offsets) return FQL_ERR_NULL_POINTER;
    if (n_slots > 0 && (!expert_of_slot || !token_of_sorted || !pos_of_slot)) return FQL_ERR_NULL_POINTER;
    const size_t lds = (size_t)(ROUTE_THREADS * E + E) * sizeof(int);
    static PerDeviceFlag attr;
    if (!ensure_lds_attr(attr, reinterpret_cast<const void *>(route_plan_kernel),
                         (ROUTE_THREADS * ROUTE_MAX_EXPERTS + ROUTE_MAX_EXPERTS) * (int)sizeof(int)))
        return FQL_ERR_LAUNCH;
    (void)hipGetLastError();
    hipLaunchKernelGGL(route_plan_kernel, dim3(1), dim3(ROUTE_THREADS), lds, static_cast<hipStream_t>(stream),
                       expert_of_slot, n_slots, top_k, E, counts, offsets, token_of_sorted, pos_of_slot);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_combine_f32(const float *y, const int32_t *pos_of_slot, const float *weights, float *out, int T, int top_k,
                    int N, int R, void *stream)
{
    if (T < 0 || top_k <= 0 || N < 0 || R < 0) return FQL_ERR_BAD_SHAPE;
    if (T == 0 || N == 0) return FQL_OK;
    if (!y || !pos_of_slot || !out || R == 0) return FQL_ERR_NULL_POINTER;      // weights == NULL: rows already weighted, pure gather-add
    if (T > 65535) return FQL_ERR_BAD_SHAPE;                 // grid.y
    hipLaunchKernelGGL(combine_kernel, dim3((N + 1023) / 1024, T), dim3(256), 0, static_cast<hipStream_t>(stream), y,
                       pos_of_slot, weights, out, T, top_k, N, R);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_regroup_index_i32(const int32_t *recv_counts, int G, int EL, int32_t *tokens_per_expert,
                          int32_t *input_offsets, int32_t *gather, int32_t *scatter, void *stream)
{
    if (G <= 0 || EL <= 0 || (size_t)G * EL > 8192) return FQL_ERR_BAD_SHAPE;
    if (!recv_counts || !tokens_per_expert || !input_offsets || !gather || !scatter) return FQL_ERR_NULL_POINTER;
    hipLaunchKernelGGL(regroup_index_kernel, dim3(1), dim3(256), (size_t)2 * G * EL * sizeof(int),
                       static_cast<hipStream_t>(stream), recv_counts, G, EL, tokens_per_expert, input_offsets, gather,
                       scatter);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_unpack_u8(const uint8_t *packed, uint8_t *q, size_t nbytes, void *stream)
{
    if (nbytes == 0) return FQL_OK;
    if (!packed || !q) return FQL_ERR_NULL_POINTER;
    size_t blocks = (nbytes / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(unpack_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       packed, q, nbytes);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_dequantize_f32(const uint8_t *packed, const float *scales, const float *zps, float *w, int N, int K,
                       void *stream)
{
    if (N < 0 || K < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (N == 0 || K == 0) return FQL_OK;
    if (!packed || !scales || !zps || !w) return FQL_ERR_NULL_POINTER;
    hipLaunchKernelGGL(dequantize_kernel, dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), packed,
                       scales, zps, w, N, K);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

size_t fql_act_limb_bytes(int T, int E, int K, int precision)
{
    const int L = limbs_of(precision);
    if (L < 0 || T <= 0 || E <= 0 || K <= 0) return 0;
    return (has_residual(L, is_f8(precision)) ? 2 : 1) * limb_bytes(L, T, E, padded_k(K));
}

size_t fql_gemm_scratch_bytes(int precision)
{
    const int L = limbs_of(precision);
    return (L >= 0 && has_residual(L, is_f8(precision))) ? res_scratch_bytes() : 0;
}

int fql_act_quant_f32(const float *x, int8_t *limbs, float *delta, int32_t *rowsum,
                      const int32_t *tokens_per_expert, const int32_t *input_offsets, int E, int T, int K,
                      int precision, void *stream)
{
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (T < 0 || K <= 0 || E <= 0) return FQL_ERR_BAD_SHAPE;
    if (T == 0) return FQL_OK;
    if (!x || !limbs || !delta || !rowsum) return FQL_ERR_NULL_POINTER;
    if ((tokens_per_expert == nullptr) != (input_offsets == nullptr)) return FQL_ERR_NULL_POINTER;
    if (tokens_per_expert == nullptr && E != 1) return FQL_ERR_BAD_SHAPE;
    if (!aligned16(limbs)) return FQL_ERR_ALIGNMENT;
    Workspace w;
    w.limbs = limbs; w.delta = delta; w.rowsum = rowsum; w.scratch = nullptr; w.bias = nullptr; w.bytes = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Kp = padded_k(K), MBT = row_blocks(T, E);
    if (L == 1) return launch_act_quant<1>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, nullptr, FQL_DTYPE_F32, 0, tokens_per_expert, input_offsets, E, st, false, is_f8(precision));
    if (L == 2) return launch_act_quant<2>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, nullptr, FQL_DTYPE_F32, 0, tokens_per_expert, input_offsets, E, st);
    return launch_act_quant<3>(x, FQL_DTYPE_F32, nullptr, 0, w, T, K, Kp, MBT, nullptr, FQL_DTYPE_F32, 0, tokens_per_expert, input_offsets, E, st);
}

int fql_quantize_rows_f32(const float *w, uint8_t *packed, float *scales, float *zps, int N, int K, void *stream)
{
    if (N < 0 || K < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (N == 0 || K == 0) return FQL_OK;
    if (!w || !packed || !scales || !zps) return FQL_ERR_NULL_POINTER;
    hipLaunchKernelGGL(quantize_rows_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), w, N, K, packed,
                       scales, zps);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

int fql_quantize_tensor_f32(const float *w, uint8_t *packed, float *scales, float *zps, float *scratch, int N, int K,
                            void *stream)
{
    if (N < 0 || K < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (N == 0 || K == 0) return FQL_OK;
    if (!w || !packed || !scales || !zps || !scratch) return FQL_ERR_NULL_POINTER;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(row_minmax_kernel, dim3(N), dim3(256), 0, st, w, N, K, scratch, scratch + N);
    hipLaunchKernelGGL(tensor_scale_kernel, dim3(1), dim3(256), 0, st, scratch, scratch + N, N, scales, zps);
    hipLaunchKernelGGL(quantize_given_kernel, dim3(N), dim3(256), 0, st, w, N, K, scales, zps, packed);
    return hipGetLastError() == hipSuccess ? FQL_OK : FQL_ERR_LAUNCH;
}

static int gemm_i8_entry(int cfg, const int8_t *limbs, const float *delta, const int32_t *rowsum,
                         const uint8_t *packed, const float *scales, const float *zps,
                         const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E, int T,
                         int K, int N, int precision, void *stream, void *scratch, size_t scratch_bytes)
{
    const int L = limbs_of(precision);
    if (L < 0) return FQL_ERR_BAD_PRECISION;
    if (E <= 0 || T < 0 || K <= 0 || N < 0) return FQL_ERR_BAD_SHAPE;
    if (K & 1) return FQL_ERR_ODD_K;
    if (T == 0 || N == 0) return FQL_OK;
    if (!limbs || !delta || !rowsum || !packed || !scales || !zps || !out) return FQL_ERR_NULL_POINTER;
    if ((tokens_per_expert == nullptr) != (input_offsets == nullptr)) return FQL_ERR_NULL_POINTER;
    if (tokens_per_expert == nullptr && E != 1) return FQL_ERR_BAD_SHAPE;
    if ((K % 32) != 0 || !aligned16(packed) || !aligned16(limbs)) return FQL_ERR_ALIGNMENT;
    if (!mfma_addressable(L, T, E, K, N)) return FQL_ERR_BAD_SHAPE;
    if (is_f8(precision)) {
        if (cfg < 0) cfg = choose_cfg_f8(E, T, N, tokens_per_expert != nullptr);
        if (!valid_cfg_f8(cfg)) return FQL_ERR_BAD_SHAPE;
    } else {
        if (cfg < 0) cfg = choose_cfg(L, E, T, K, N, tokens_per_expert != nullptr);
        if (!valid_cfg(cfg, L)) return FQL_ERR_BAD_SHAPE;
    }
    Workspace w;
    w.limbs = const_cast<int8_t *>(limbs);
    w.delta = const_cast<float *>(delta);
    w.rowsum = const_cast<int32_t *>(rowsum);
    w.bias = nullptr;
    w.row_weight = nullptr;
    w.bytes = 0;
    // without (enough) scratch the residual pass of heavy-tailed rows is skipped: the result is then the plain 8L-1 bit one
    // without (enough) scratch the residual pass of heavy-tailed rows is skipped: the result is then the plain 8L-1 bit one
    w.scratch = (has_residual(L, is_f8(precision)) && scratch != nullptr && aligned16(scratch) && scratch_bytes >= res_scratch_bytes())
                    ? static_cast<float *>(scratch) : nullptr;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int Kp = padded_k(K), MBT = row_blocks(T, E);
    if (is_f8(precision))
        return launch_gemm_f8(cfg, w, packed, scales, zps, out, FQL_DTYPE_F32, tokens_per_expert, input_offsets, E, T, K, Kp, MBT, N, st);
    if (L == 1)
        return launch_gemm<1>(cfg, w, packed, scales, zps, out, FQL_DTYPE_F32, tokens_per_expert, input_offsets, E, T, K, Kp, MBT, N, st);
    if (L == 2)
        return launch_gemm<2>(cfg, w, packed, scales, zps, out, FQL_DTYPE_F32, tokens_per_expert, input_offsets, E, T, K, Kp, MBT, N, st);
    return launch_gemm<3>(cfg, w, packed, scales, zps, out, FQL_DTYPE_F32, tokens_per_expert, input_offsets, E, T, K, Kp, MBT, N, st);
}

int fql_gemm_i8_f32(const int8_t *limbs, const float *delta, const int32_t *rowsum, const uint8_t *packed,
                    const float *scales, const float *zps, const int32_t *tokens_per_expert,
                    const int32_t *input_offsets, float *out, int E, int T, int K, int N, int precision,
                    void *stream, void *scratch, size_t scratch_bytes)
{
    return gemm_i8_entry(-1, limbs, delta, rowsum, packed, scales, zps, tokens_per_expert, input_offsets, out, E, T,
                         K, N, precision, stream, scratch, scratch_bytes);
}

// Tuning hooks (include/fql_int4_tune.h, not part of the drop-in boundary): the same call with an explicit tile configuration id.
FQL_API int fql_tune_gemm_i8_f32(int cfg, const int8_t *limbs, const float *delta, const int32_t *rowsum,
                                 const uint8_t *packed, const float *scales, const float *zps,
                                 const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E,
                                 int T, int K, int N, int precision, void *stream, void *scratch, size_t scratch_bytes)
{
    if (!(precision == FQL_PRECISION_FP8 ? valid_cfg_f8(cfg) : valid_cfg(cfg, limbs_of(precision)))) return FQL_ERR_BAD_SHAPE;
    return gemm_i8_entry(cfg, limbs, delta, rowsum, packed, scales, zps, tokens_per_expert, input_offsets, out, E, T,
                         K, N, precision, stream, scratch, scratch_bytes);
}

#if defined(FQL_TRACE)
FQL_API int fql_debug_trace_act(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fql_trace_act), sizeof(unsigned long long) * 16 * 16) == hipSuccess ? 0 : -1;
}
FQL_API int fql_debug_trace_wide(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fql_trace_wide), sizeof(unsigned long long) * 8 * 64) == hipSuccess ? 0 : -1;
}
FQL_API int fql_debug_trace(unsigned long long *dst)
{
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(fql_trace_buf), sizeof(unsigned long long) * 8 * 64) == hipSuccess ? 0 : -1;
}
#endif
FQL_API int fql_tune_set_act_single_rows(int rows) { const int old = g_act_single_rows; if (rows >= 0) g_act_single_rows = rows; return old; }
FQL_API int fql_tune_set_balance_tiles(int on) { const int old = g_balance_tiles; g_balance_tiles = on ? 1 : 0; return old; }
FQL_API int fql_tune_set_group_mfma(int on) { const int old = g_group_mfma; g_group_mfma = on ? 1 : 0; return old; }
FQL_API int fql_tune_set_group_i8_min_rows(int rows) { const int old = g_group_i8_min_rows; if (rows >= 1) { g_group_i8_min_rows = rows; g_group_i8_min_rows_grouped = rows; } return old; }
FQL_API int fql_tune_set_group_i8(int on) { const int old = g_group_i8; g_group_i8 = on ? 1 : 0; return old; }
FQL_API int fql_tune_set_gemv_max_rows(int rows) { const int old = g_gemv_max_rows; if (rows >= 0 && rows <= 4) g_gemv_max_rows = rows; return old; }
FQL_API int fql_tune_num_configs(void) { return FQL_NUM_CFG; }
FQL_API int fql_tune_is_config(int cfg, int precision)
{
    const int L = limbs_of(precision);
    if (L < 0) return 0;
    return (is_f8(precision) ? valid_cfg_f8(cfg) : valid_cfg(cfg, L)) ? 1 : 0;
}
FQL_API int fql_tune_num_rows32_configs(void) { return FQL_NUM_ROWS32; }
FQL_API int fql_tune_num_rows16_configs(void) { return FQL_NUM_ROWS16; }
FQL_API int fql_tune_num_w4_configs(void) { return FQL_NUM_W4; }
// which tile configuration the product path picks for a shape (ids as in fql_tune_gemm_i8_f32; bench.py labels its roofline with it)
FQL_API int fql_tune_chosen_cfg(int precision, int E, int T, int K, int N, int grouped)
{
    const int L = limbs_of(precision);
    if (L < 0) return -1;
    return is_f8(precision) ? choose_cfg_f8(E, T, N, grouped != 0) : choose_cfg(L, E, T, K, N, grouped != 0);
}
FQL_API int fql_tune_set_compute_units(int n) { const int old = g_cu_cap; g_cu_cap = n > 0 ? (n < 8 ? 8 : n - n % 8) : 0; return old; }
FQL_API int fql_tune_set_fused(int on) { const int old = g_fused; g_fused = on ? 1 : 0; return old; }
FQL_API int fql_tune_set_fused_spin(int polls) { const int old = g_fused_spin; if (polls >= 0) g_fused_spin = polls; return old; }
FQL_API int fql_tune_set_w4(int on) { const int old = g_use_w4; g_use_w4 = on ? 1 : 0; return old; }


}  // extern "C"
