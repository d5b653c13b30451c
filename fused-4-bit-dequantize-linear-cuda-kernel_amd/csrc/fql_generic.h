// Shape-generic fallback and format helpers (any even K, no alignment assumptions).
//
// fused_rows_kernel: float32 FMA formulation of the reference kernel
// (csrc/quantized_linear_kernel.cu:218-264: dq = (q - zp) * scale; sum = fma(dq, x, sum)) with the
// work laid out for a 64-wide wavefront: one wave per output row n, lanes stride over the packed
// bytes of that row (coalesced), up to RB batch rows share every weight byte, wave-level reduce.
// Used when K % 32 != 0 or a base pointer is not 16-byte aligned; the aligned shapes take the
// GEMV / MFMA kernels instead.
#pragma once
#include "fql_common.h"

template <int RB>
__global__ __launch_bounds__(256) void fused_rows_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ packed, const float *__restrict__ scales,
    const float *__restrict__ zps, float *__restrict__ out, const int32_t *__restrict__ tpe,
    const int32_t *__restrict__ offs, int T, int K, int N, const float *__restrict__ bias)
{
    const int e = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    int row_lo = 0, row_hi = T;
    if (tpe != nullptr) {
        long long lo = offs[e], hi = lo + (long long)tpe[e];
        row_lo = (int)(lo < 0 ? 0 : lo);
        row_hi = (int)(hi > T ? T : hi);
    }
    const int K2 = K >> 1;
    const uint8_t *prow = packed + ((size_t)e * N + n) * K2;
    const float sc = scales[(size_t)e * N + n];
    const float zp = zps[(size_t)e * N + n];
    for (int b0 = row_lo; b0 < row_hi; b0 += RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
        for (int j = lane; j < K2; j += 64) {
            const uint8_t byte = prow[j];
            const float w0 = ((float)(byte & 0x0F) - zp) * sc;
            const float w1 = ((float)(byte >> 4) - zp) * sc;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (b0 + r < row_hi) {
                    const float *xr = x + (size_t)(b0 + r) * K;
                    acc[r] = fmaf(w0, xr[2 * j], acc[r]);
                    acc[r] = fmaf(w1, xr[2 * j + 1], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float s = wave_sum(acc[r]);
            if (lane == 0 && b0 + r < row_hi) out[(size_t)(b0 + r) * N + n] = bias != nullptr ? s + bias[(size_t)e * N + n] : s;
        }
    }
}

// Per-GROUP scales and zero points along K (not in the reference, whose quantisation is per row:
// python/quantize.py:73-80): scales / zps are [E][N][K / group] and
//   out[t][n] = sum_k x[t][k] * (q[n][k] - zp[n][k / group]) * scale[n][k / group]        (float32 FMA, as above).
// GPTQ / AWQ-style checkpoints: the same one-wave-per-output-row kernel, the group's two constants looked up per packed
// byte (L1 hits).  Takes the calls with fewer than 4 rows per group and the shapes fql_group.h (the float32
// matrix-core kernel for batches) does not: K % 64 != 0, group % 32 != 0, unaligned bases.
template <int RB>
__global__ __launch_bounds__(256) void fused_rows_group_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ packed, const float *__restrict__ scales,
    const float *__restrict__ zps, float *__restrict__ out, const int32_t *__restrict__ tpe,
    const int32_t *__restrict__ offs, int T, int K, int N, int group, const float *__restrict__ bias)
{
    const int e = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    int row_lo = 0, row_hi = T;
    if (tpe != nullptr) {
        long long lo = offs[e], hi = lo + (long long)tpe[e];
        row_lo = (int)(lo < 0 ? 0 : lo);
        row_hi = (int)(hi > T ? T : hi);
    }
    const int K2 = K >> 1, G = K / group;
    const uint8_t *prow = packed + ((size_t)e * N + n) * K2;
    const float *srow = scales + ((size_t)e * N + n) * G;
    const float *zrow = zps + ((size_t)e * N + n) * G;
    for (int b0 = row_lo; b0 < row_hi; b0 += RB) {
        float acc[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = 0.0f;
        for (int j = lane; j < K2; j += 64) {
            const uint8_t byte = prow[j];
            const int g = (2 * j) / group;                    // group is even: both nibbles of a byte share it
            const float sc = srow[g], zp = zrow[g];
            const float w0 = ((float)(byte & 0x0F) - zp) * sc;
            const float w1 = ((float)(byte >> 4) - zp) * sc;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                if (b0 + r < row_hi) {
                    const float *xr = x + (size_t)(b0 + r) * K;
                    acc[r] = fmaf(w0, xr[2 * j], acc[r]);
                    acc[r] = fmaf(w1, xr[2 * j + 1], acc[r]);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const float s = wave_sum(acc[r]);
            if (lane == 0 && b0 + r < row_hi) out[(size_t)(b0 + r) * N + n] = bias != nullptr ? s + bias[(size_t)e * N + n] : s;
        }
    }
}

// Zero the rows of out[T][N] that no expert range covers (torch::zeros semantics of the reference's
// MoE wrapper, csrc/moe_int4_kernel.cu:109).  Only the generic path launches this; the MFMA path
// folds it into the activation pre-pass.
__global__ __launch_bounds__(256) void zero_uncovered_rows_kernel(
    float *__restrict__ out, const int32_t *__restrict__ tpe, const int32_t *__restrict__ offs,
    int E, int T, int N)
{
    const int t = blockIdx.x;
    bool covered = false;
    for (int e = 0; e < E; ++e) {
        long long lo = offs[e], hi = lo + (long long)tpe[e];
        lo = lo < 0 ? 0 : lo;
        hi = hi > T ? T : hi;
        covered |= (t >= lo && t < hi);
    }
    if (covered) return;
    for (int i = threadIdx.x; i < N; i += 256) out[(size_t)t * N + i] = 0.0f;
}

// q[2j] = byte & 15, q[2j+1] = byte >> 4   (python/quantize.py:152-163)
__global__ __launch_bounds__(256) void unpack_u8_kernel(const uint8_t *__restrict__ packed,
                                                        uint8_t *__restrict__ q, size_t nbytes)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < nbytes; i += stride) {
        if (i + 4 <= nbytes && ((reinterpret_cast<uintptr_t>(packed) | reinterpret_cast<uintptr_t>(q)) & 7) == 0) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(packed + i);
            uint32_t lo, hi;
            unpack8(w, lo, hi);                       // lo: even k, hi: odd k, one per byte
            // interleave back to natural k order: bytes (lo0,hi0,lo1,hi1 | lo2,hi2,lo3,hi3)
            const uint32_t o0 = __builtin_amdgcn_perm(hi, lo, 0x05010400u);
            const uint32_t o1 = __builtin_amdgcn_perm(hi, lo, 0x07030602u);
            *reinterpret_cast<uint2 *>(q + 2 * i) = make_uint2(o0, o1);
        } else {
            for (size_t j = i; j < nbytes && j < i + 4; ++j) {
                q[2 * j] = packed[j] & 0x0F;
                q[2 * j + 1] = packed[j] >> 4;
            }
        }
    }
}

// w[n][k] = (q - zp[n]) * scale[n]   (python/quantize.py:172); one wave per row, coalesced.
__global__ __launch_bounds__(256) void dequantize_kernel(
    const uint8_t *__restrict__ packed, const float *__restrict__ scales, const float *__restrict__ zps,
    float *__restrict__ w, int N, int K)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int lane = threadIdx.x & 63;
    const int K2 = K >> 1;
    const float sc = scales[n], zp = zps[n];
    const uint8_t *prow = packed + (size_t)n * K2;
    float *wrow = w + (size_t)n * K;
    for (int j = lane; j < K2; j += 64) {
        const uint8_t b = prow[j];
        wrow[2 * j] = ((float)(b & 0x0F) - zp) * sc;
        wrow[2 * j + 1] = ((float)(b >> 4) - zp) * sc;
    }
}
