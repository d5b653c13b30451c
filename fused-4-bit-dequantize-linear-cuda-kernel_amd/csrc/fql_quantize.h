// GPU quantisers: the construction-time half of the weight format, bit-exact with the
// reference's CPU arithmetic so that weights quantised on the GPU equal weights quantised on the host.
//
//   quantize_rows_kernel    python/quantize.py:38-124       per-ROW asymmetric INT4 + nibble packing
//   tensor_minmax / quantize_uniform  python/moe_int4_module.py:45-76   per-TENSOR (per-expert) variant
//
// HBM-bound: read N*K*4 bytes (second pass from L2), write N*K/2.  One 256-thread workgroup per row.
// Every float operation is a single IEEE operation in the reference's order (hipcc's default
// correctly-rounded division, no fast-math, rintf = round-half-to-even = torch.round).
#pragma once
#include "fql_common.h"
#include <math.h>

__device__ __forceinline__ float block_reduce_minmax(float v, bool is_max, float *sh)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float n = __shfl_xor(v, o, 64);
        v = is_max ? fmaxf(v, n) : fminf(v, n);
    }
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < 4; ++i) r = is_max ? fmaxf(r, sh[i]) : fminf(r, sh[i]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ uint8_t quant_pair(float w0, float w1, float scale, float zp)
{
    float q0 = rintf(w0 / scale + zp), q1 = rintf(w1 / scale + zp);      // :106-109
    q0 = fminf(fmaxf(q0, 0.0f), 15.0f);
    q1 = fminf(fmaxf(q1, 0.0f), 15.0f);
    return (uint8_t)(((uint32_t)q1 << 4) | (uint32_t)q0);                // :120-122
}

// weight [N][K] f32 -> packed [N][K/2], scales [N], zps [N]
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float *__restrict__ w, int N, int K,
                                                            uint8_t *__restrict__ packed, float *__restrict__ scales,
                                                            float *__restrict__ zps)
{
    __shared__ float sh[4];
    const int n = blockIdx.x;
    const float *row = w + (size_t)n * K;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = row[k];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = block_reduce_minmax(mn, false, sh);
    mx = block_reduce_minmax(mx, true, sh);
    float scale = (mx - mn) / 15.0f;                                      // :80
    if (mx == mn) scale = fmaxf(fabsf(mx), 1.0f) / 15.0f;                 // :85-92
    scale = fmaxf(scale, 1e-8f);                                          // :94
    const float zp = fminf(fmaxf(rintf(-mn / scale), 0.0f), 15.0f);       // :100-101
    if (threadIdx.x == 0) { scales[n] = scale; zps[n] = zp; }
    uint8_t *prow = packed + (size_t)n * (K >> 1);
    for (int j = threadIdx.x; j < (K >> 1); j += 256) prow[j] = quant_pair(row[2 * j], row[2 * j + 1], scale, zp);
}

// per-row min / max of a [N][K] tensor (first half of the per-tensor quantiser)
__global__ __launch_bounds__(256) void row_minmax_kernel(const float *__restrict__ w, int N, int K,
                                                         float *__restrict__ mins, float *__restrict__ maxs)
{
    __shared__ float sh[4];
    const int n = blockIdx.x;
    const float *row = w + (size_t)n * K;
    float mn = INFINITY, mx = -INFINITY;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float v = row[k];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = block_reduce_minmax(mn, false, sh);
    mx = block_reduce_minmax(mx, true, sh);
    if (threadIdx.x == 0) { mins[n] = mn; maxs[n] = mx; }
}

// one workgroup: reduce the row extrema, derive the tensor's scale / zero-point (no zero-range guard, as in
// the reference: python/moe_int4_module.py:49-51), broadcast them to the [N] outputs
__global__ __launch_bounds__(256) void tensor_scale_kernel(const float *__restrict__ mins, const float *__restrict__ maxs,
                                                           int N, float *__restrict__ scales, float *__restrict__ zps)
{
    __shared__ float sh[4];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < N; i += 256) { mn = fminf(mn, mins[i]); mx = fmaxf(mx, maxs[i]); }
    mn = block_reduce_minmax(mn, false, sh);
    mx = block_reduce_minmax(mx, true, sh);
    const float scale = (mx - mn) / 15.0f;
    const float zp = fminf(fmaxf(rintf(-mn / scale), 0.0f), 15.0f);
    for (int i = threadIdx.x; i < N; i += 256) { scales[i] = scale; zps[i] = zp; }
}

// quantise + pack with per-row scale / zero-point already on the device
__global__ __launch_bounds__(256) void quantize_given_kernel(const float *__restrict__ w, int N, int K,
                                                             const float *__restrict__ scales,
                                                             const float *__restrict__ zps, uint8_t *__restrict__ packed)
{
    const int n = blockIdx.x;
    const float *row = w + (size_t)n * K;
    const float scale = scales[n], zp = zps[n];
    uint8_t *prow = packed + (size_t)n * (K >> 1);
    for (int j = threadIdx.x; j < (K >> 1); j += 256) prow[j] = quant_pair(row[2 * j], row[2 * j + 1], scale, zp);
}
