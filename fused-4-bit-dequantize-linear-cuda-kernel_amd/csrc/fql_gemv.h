// Small-batch (B <= 4) fused INT4 dequantize-linear: the HBM-bound regime (4 flop per weight byte
// at B = 1), so the kernel is organised around streaming the packed weights exactly once at full
// width and keeping everything else on chip.
//
//   * each lane loads 16 packed bytes (32 weights) per row with one global_load_dwordx4; a wave
//     covers 1 KiB contiguous of a weight row per instruction, 4 rows in flight per wave.
//   * the activations (B x K float32) are staged ONCE per workgroup into LDS, each 32-float
//     segment padded to 36 floats so that the per-lane ds_read_b128 of "my 32 k" is conflict-free.
//   * nibbles -> float by mask/shift + v_cvt_f32_ubyteN (11 VALU per 8 weights), float32 FMA,
//     out = scale * (sum_k q*x - zp * sum_k x): the zero-point is folded through the row sum of x
//     so the inner loop has no per-weight subtract.
//   * wave-level shuffle reduction, one float store per (row, b).
//
// Algorithmic bytes per call (the reference's own model, benchmark/run_benchmark.py:222):
//   N*K/2 + 8*N + 4*B*K read, 4*B*N written.
#pragma once
#include "fql_common.h"

#define GEMV_ROWS 4          // weight rows in flight per wave
#define GEMV_SEG 36          // floats per padded 32-float LDS segment

// GROUPED: per-GROUP scales along K (scales / zps [N][K / group], group % 32 == 0): a lane's 32 k of
// a chunk share one group, so the chunk's dot product and the chunk's sum of x are folded with that group's constants,
//   acc += scale[n][g] * (sum_k q x - zp[n][g] * sum_k x),
// instead of once per row at the end.
template <int B, bool GROUPED = false>
__global__ __launch_bounds__(256) void gemv_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ packed, const float *__restrict__ scales,
    const float *__restrict__ zps, float *__restrict__ out, int K, int N, const float *__restrict__ bias, int group = 0)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];      // [B][K/32][36] + sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int segs = K >> 5;
    float *sumx = xs + (size_t)B * segs * GEMV_SEG;                  // [B][4 waves]

    // ---- stage x into LDS (coalesced float4), and its row sums
    float part[B];
#pragma unroll
    for (int b = 0; b < B; ++b) part[b] = 0.0f;
    for (int i = tid; i < (K >> 2); i += 256) {
        const int seg = i >> 3, q4 = i & 7;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const v4f v = *reinterpret_cast<const v4f *>(x + (size_t)b * K + 4 * i);
            *reinterpret_cast<v4f *>(xs + ((size_t)b * segs + seg) * GEMV_SEG + 4 * q4) = v;
            part[b] += (v[0] + v[1]) + (v[2] + v[3]);
        }
    }
#pragma unroll
    for (int b = 0; b < B; ++b) {
        const float s = wave_sum(part[b]);
        if (lane == 0) sumx[b * 4 + wave] = s;
    }
    __syncthreads();
    float sx[B];
#pragma unroll
    for (int b = 0; b < B; ++b) sx[b] = (sumx[b * 4] + sumx[b * 4 + 1]) + (sumx[b * 4 + 2] + sumx[b * 4 + 3]);

    const int K2 = K >> 1;                 // packed bytes per row
    const int nvec = K2 >> 4;              // uint4 per row
    const int groups = (N + GEMV_ROWS - 1) / GEMV_ROWS;
    for (int rg = blockIdx.x * 4 + wave; rg < groups; rg += gridDim.x * 4) {
        const int n0 = rg * GEMV_ROWS;
        float acc[GEMV_ROWS][B];
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) acc[r][b] = 0.0f;

        for (int j = lane; j < nvec; j += 64) {
            uint4 w[GEMV_ROWS];
#pragma unroll
            for (int r = 0; r < GEMV_ROWS; ++r) {
                const int n = (n0 + r < N) ? n0 + r : N - 1;
                w[r] = *reinterpret_cast<const uint4 *>(packed + (size_t)n * K2 + (size_t)j * 16);
            }
            float gsc[GEMV_ROWS], gzp[GEMV_ROWS];
            if constexpr (GROUPED) {
                const int G = K / group, g = (32 * j) / group;
#pragma unroll
                for (int r = 0; r < GEMV_ROWS; ++r) {
                    const int n = (n0 + r < N) ? n0 + r : N - 1;
                    gsc[r] = scales[(size_t)n * G + g];
                    gzp[r] = zps[(size_t)n * G + g];
                }
            }
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float *xb = xs + ((size_t)b * segs + j) * GEMV_SEG;
                v4f xv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const v4f *>(xb + 4 * i);
                float xsum = 0.0f;
                if constexpr (GROUPED) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) xsum += (xv[i][0] + xv[i][1]) + (xv[i][2] + xv[i][3]);
                }
#pragma unroll
                for (int r = 0; r < GEMV_ROWS; ++r) {
                    const uint32_t ww[4] = {w[r].x, w[r].y, w[r].z, w[r].w};
                    float a = GROUPED ? 0.0f : acc[r][b];
#pragma unroll
                    for (int d = 0; d < 4; ++d) {           // dword d holds k = 8d .. 8d+7
                        uint32_t lo, hi;
                        unpack8(ww[d], lo, hi);
#pragma unroll
                        for (int i = 0; i < 4; ++i) {       // byte i: lo -> k = 2i, hi -> k = 2i+1
                            const float q0 = (float)((lo >> (8 * i)) & 0xFFu);
                            const float q1 = (float)((hi >> (8 * i)) & 0xFFu);
                            const int k = 8 * d + 2 * i;
                            a = fmaf(q0, xv[k >> 2][k & 3], a);
                            a = fmaf(q1, xv[(k + 1) >> 2][(k + 1) & 3], a);
                        }
                    }
                    if constexpr (GROUPED) acc[r][b] = fmaf(gsc[r], fmaf(-gzp[r], xsum, a), acc[r][b]);
                    else acc[r][b] = a;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < GEMV_ROWS; ++r) {
            const int n = n0 + r;
            const float sc = (!GROUPED && n < N) ? scales[n] : 0.0f;
            const float zp = (!GROUPED && n < N) ? zps[n] : 0.0f;
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float dot = wave_sum(acc[r][b]);
                if (lane == 0 && n < N) {
#pragma clang fp contract(off)                               // the bias is ONE float32 add after the rounded un-biased result
                    const float v = GROUPED ? dot : sc * fmaf(-zp, sx[b], dot);
                    out[(size_t)b * N + n] = bias != nullptr ? v + bias[n] : v;
                }
            }
        }
    }
}

