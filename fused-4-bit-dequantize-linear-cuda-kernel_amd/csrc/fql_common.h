// Shared device helpers for libfql_int4 (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define FQL_WAVE 64
// K-depth of one weight stage of the MFMA GEMM: 256 k = 128 packed bytes = one full cache line per
// weight row.  The activation limbs are written by the pre-pass in blocks of the same depth.
#define FQL_KB 256
// Rows of one MFMA row-block (v_mfma_i32_32x32x32_i8).  Every expert's rows start at a multiple of
// this in the limb workspace, so a wave's A fragment is one contiguous 1 KiB.
#define FQL_MB 32

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
