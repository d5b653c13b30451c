// Shared device helpers for libfql_int4 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define FQL_WAVE 64
// K-depth of one weight stage of the MFMA GEMM: 256 k = 128 packed bytes = one full cache line per
// weight row.  The activation limbs are written by the pre-pass in blocks of the same depth.
#define FQL_KB 256
// Rows of one MFMA row-block (v_mfma_i32_32x32x32_i8).  Every expert's rows start at a multiple of
// this in the limb workspace, so a wave's A fragment is one contiguous 1 KiB.
#define FQL_MB 32

// E8M0 block scales of the scaled matrix-core instruction, one byte per lane (op_sel 0 = byte 0): 2^(v - 127)
#define FQL_E8M0_ONE 0x7F7F7F7F
#define FQL_E8M0_2P9 0x88888888

#ifndef FQL_RES_ENABLED
#define FQL_RES_ENABLED 1      // residual pass of heavy-tailed rows in the GEMM kernels (0: A/B timing builds only)
#endif

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))

// Nibble unpack shared by every kernel (reference: python/quantize.py:152-163,
// csrc/quantized_linear_kernel.cu:223-224).  One packed dword holds k0..k7 with byte j =
// q[2j] | q[2j+1] << 4.  `lo` gets (k0,k2,k4,k6), `hi` gets (k1,k3,k5,k7), one value per byte.
__device__ __forceinline__ void unpack8(uint32_t w, uint32_t &lo, uint32_t &hi)
{
    lo = w & 0x0F0F0F0Fu;
    hi = (w >> 4) & 0x0F0F0F0Fu;
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- output element types (FQL_DTYPE_* of include/fql_int4.h): 0 float32, 1 float16, 2 bfloat16, both rounded
//      to nearest even exactly as torch's .to(dtype) does
__device__ __forceinline__ unsigned short f32_to_f16_bits(float f)
{
    const _Float16 h = (_Float16)f;
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f)
{
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x40u);     // quiet NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// Write-through (sc1) 16-byte store: the line goes straight to memory and is dropped from the XCD's L2, so nothing of
// it is left dirty for the kernel-boundary write-back (MI355X_MICROARCH.md, "boundary": + bytes / 6 TB/s for what the
// predecessor leaves dirty).  Used for data this kernel never reads again.  The s_nop keeps the data registers intact
// until the store has read them (cdna_hip_programming.md 5.7 item 1).
#ifndef FQL_OUT_WT
#define FQL_OUT_WT 0           // GEMM outputs write-through (A/B knob)
#endif
#ifndef FQL_LIMB_WT
#define FQL_LIMB_WT 0          // pre-pass limb stores write-through (A/B knob)
#endif
__device__ __forceinline__ void store16_wt(void *p, v4f v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store16_wt(void *p, v4i v)
{
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// Four consecutive outputs of row `row_elems` (= t * N) starting at column n.  `vec`: N % 4 == 0 and a suitably
// aligned base, so the four columns exist and one wide store is legal.
__device__ __forceinline__ void store_out4(void *out, int kind, size_t row_elems, int n, int N, bool vec, const float (&o)[4])
{
    if (kind == 0) {
        float *p = reinterpret_cast<float *>(out) + row_elems + n;
        if (vec) {
            if (n < N) {
                if (FQL_OUT_WT) store16_wt(p, v4f{o[0], o[1], o[2], o[3]});
                else *reinterpret_cast<v4f *>(p) = v4f{o[0], o[1], o[2], o[3]};
            }
        }
        else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n + c < N) p[c] = o[c];
        }
    } else {
        unsigned short h[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) h[c] = (kind == 1) ? f32_to_f16_bits(o[c]) : f32_to_bf16_bits(o[c]);
        unsigned short *p = reinterpret_cast<unsigned short *>(out) + row_elems + n;
        if (vec) {
            if (n < N) *reinterpret_cast<uint2 *>(p) = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) if (n + c < N) p[c] = h[c];
        }
    }
}

// Bijective XCD-aware remap of a 1-D grid: blocks b and b+8 share an XCD (round-robin dispatch),
// so give each XCD a contiguous range of logical tile ids.  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nblk)
{
    const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    const int base = (x < r) ? x * (q + 1) : r * (q + 1) + (x - r) * q;
    return base + (bid >> 3);
}

// Expert row ranges live on the device (tokens_per_expert / input_offsets).  Clip to [0, T].
__device__ __forceinline__ void expert_range(const int32_t *tpe, const int32_t *offs, int i, int T, int &lo, int &cnt)
{
    long long a = offs[i], b = a + (long long)tpe[i];
    a = a < 0 ? 0 : a;
    b = b > T ? T : b;
    lo = (int)a;
    cnt = b > a ? (int)(b - a) : 0;
}

// ---- expert table by one wavefront: lane i owns expert base + i.  All E (lo, cnt) pairs are fetched with
//      ONE vector load per array (a sequential scalar scan costs ~2 dependent L2 round trips per expert at
//      the head of every workgroup) and the running sums come from a 6-step shuffle scan.
struct ExpertLane {
    int lo, cnt;        // clipped row range of this lane's expert (cnt = 0 for lanes past E)
    int pad_excl;       // padded rows (multiples of FQL_MB) of all earlier experts
    int tile_excl;      // m-tiles of all earlier experts (tile height bm)
    int tiles;          // m-tiles of this expert
};

// Inclusive prefix sum over the 64 lanes on the DPP data path: shifts inside the 16-lane rows, then the last lane of a
// row broadcast into the next row(s) -- 6 adds of ~8 cycles each.  (The shuffle form, 6 dependent ds_bpermutes through
// the LDS crossbar, cost ~100 cycles a step, and every persistent kernel runs four to nine of these scans before its
// first load: 1.4 us of the GEMM's prologue on the in-kernel clock, tools/trace_step.py.)
__device__ __forceinline__ int wave_incl_scan(int v, int lane)
{
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1  (lanes without a source add 0)
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}
// The value lane `src` holds, for a wave-uniform src: v_readlane instead of a trip through the LDS crossbar.
__device__ __forceinline__ int wave_bcast(int v, int src)
{
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}

// Processes experts [base, base + 64); carry_pad / carry_tile are the totals of the experts before `base`
// and are advanced to include this chunk.  In two halves, so that a caller can put other loads between the table's loads
// and their first use: expert_chunk_load issues them (raw offset / count of this lane's expert, 0 / 0 past E).
__device__ __forceinline__ void expert_chunk_load(const int32_t *tpe, const int32_t *offs, int E, int base, int lane, int &off_raw, int &cnt_raw)
{
    off_raw = 0; cnt_raw = 0;
    if (base + lane < E) { off_raw = offs[base + lane]; cnt_raw = tpe[base + lane]; }
}
__device__ __forceinline__ ExpertLane expert_chunk_scan(int off_raw, int cnt_raw, int T, int bm, int lane, int &carry_pad, int &carry_tile)
{
    ExpertLane r;
    {
        long long a = off_raw, b = a + (long long)cnt_raw;   // (the clipping of expert_range)
        a = a < 0 ? 0 : a;
        b = b > T ? T : b;
        r.lo = (int)a;
        r.cnt = b > a ? (int)(b - a) : 0;
    }
    const int pad = (r.cnt + FQL_MB - 1) / FQL_MB * FQL_MB;
    r.tiles = (r.cnt + bm - 1) / bm;
    const int pad_incl = wave_incl_scan(pad, lane);
    const int tile_incl = wave_incl_scan(r.tiles, lane);
    r.pad_excl = carry_pad + pad_incl - pad;
    r.tile_excl = carry_tile + tile_incl - r.tiles;
    carry_pad += wave_bcast(pad_incl, 63);
    carry_tile += wave_bcast(tile_incl, 63);
    return r;
}
__device__ __forceinline__ ExpertLane expert_chunk(const int32_t *tpe, const int32_t *offs, int E, int T, int bm,
                                                   int base, int lane, int &carry_pad, int &carry_tile)
{
    ExpertLane r;
    r.lo = 0; r.cnt = 0;
    if (base + lane < E) expert_range(tpe, offs, base + lane, T, r.lo, r.cnt);
    const int pad = (r.cnt + FQL_MB - 1) / FQL_MB * FQL_MB;
    r.tiles = (r.cnt + bm - 1) / bm;
    const int pad_incl = wave_incl_scan(pad, lane);
    const int tile_incl = wave_incl_scan(r.tiles, lane);
    r.pad_excl = carry_pad + pad_incl - pad;
    r.tile_excl = carry_tile + tile_incl - r.tiles;
    carry_pad += wave_bcast(pad_incl, 63);
    carry_tile += wave_bcast(tile_incl, 63);
    return r;
}
