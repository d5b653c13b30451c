// Per-GROUP scales along K on the matrix cores (not in the reference, whose quantisation is per row:
// python/quantize.py:73-80): scales / zps are [E][N][K / group] and
//   out[t][n] = sum_k x[t][k] * (q[n][k] - zp[n][k / group]) * scale[n][k / group].
// The integer kernels cannot carry this (they need ONE scale per output row to keep their accumulators integer across
// K), so the weights are dequantised in registers, exactly as the reference kernel does per element
// (csrc/quantized_linear_kernel.cu:240-244: dq = (q - zp) * scale), and the contraction runs on the float32 matrix-core
// instruction v_mfma_f32_32x32x2f32: float32 products, float32 accumulation -- the arithmetic of the reference's FMA
// chain in another summation order.  157 TFLOP/s is the ceiling of that instruction (1/32 of the INT8 rate): this is the
// batch path of an additive option, not the headline; batches of a few rows stay on the one-wave-per-output-row kernel
// (fql_generic.h), which streams the weights with less padding.
//
// Workgroup = 4 waves = a 64 or 128 (n) x 64 (t) tile: one or two 32 x 32 blocks per wave (two share the activation
// registers; the host takes them when that still leaves every CU a few tiles).  Per 64-k chunk a lane of half h = lane >> 5
// owns the 32 consecutive k = k0 + 32 h + [0, 32): 16 packed bytes of weight row n = lane & 31 (one 16-byte load) and
// 32 floats of activation row t = lane & 31 (eight 16-byte loads), and issues 32 MFMAs, each contracting one k of
// either half.  Which two k an instruction contracts is free as long as both operands agree.  group % 32 == 0 keeps a
// lane's 32 k inside one group.
#pragma once
#include "fql_common.h"

// KSPLIT (a few dozen rows per group: the op is a weight stream and a 64 x 64 tile per workgroup leaves CUs idle and
// waves padding): the workgroup is ONE 32 x 32 block and its four waves take a quarter of K each, summed through LDS.
template <bool KSPLIT, int NBLK = 1>
__global__ __launch_bounds__(256) void group_mfma_kernel(
    const float *__restrict__ x, const uint8_t *__restrict__ packed, const float *__restrict__ scales,
    const float *__restrict__ zps, float *__restrict__ out, const int32_t *__restrict__ tpe,
    const int32_t *__restrict__ offs, int T, int K, int N, int group, const float *__restrict__ bias)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int e = blockIdx.z;
    int row_lo = 0, row_hi = T;
    if (tpe != nullptr) {
        int lo, cnt;
        expert_range(tpe, offs, e, T, lo, cnt);
        row_lo = lo;
        row_hi = lo + cnt;
    }
    constexpr int TB = KSPLIT ? 32 : 64;                           // rows of the tile
    constexpr int NB = KSPLIT ? 1 : NBLK;                          // 32-column blocks per wave (they share the activation registers)
    constexpr int TN = KSPLIT ? 32 : 64 * NB;                      // columns of the tile
    const int t_blk = row_lo + (int)blockIdx.y * TB;
    if (t_blk >= row_hi) return;                                   // (uniform per workgroup)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int n_wave = (int)blockIdx.x * TN + (KSPLIT ? 0 : (wave & 1) * 32);   // block b of the wave: + 64 b
    const int t = t_blk + (KSPLIT ? 0 : (wave >> 1) * 32) + l31;   // this lane's activation row (B operand)
    const int k_lo = KSPLIT ? wave * (K >> 2) : 0, k_hi = KSPLIT ? k_lo + (K >> 2) : K;   // (host: K % 256 == 0)
    const int tc = t < row_hi ? t : row_hi - 1;
    const int K2 = K >> 1, G = K / group;
    const uint8_t *wrow[NB];
    const float *srow[NB], *zrow[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int n = n_wave + 64 * b + l31;                       // this lane's weight row (A operand) of block b
        const int nc = n < N ? n : N - 1;
        wrow[b] = packed + ((size_t)e * N + nc) * K2 + 16 * h;
        srow[b] = scales + ((size_t)e * N + nc) * G;
        zrow[b] = zps + ((size_t)e * N + nc) * G;
    }
    const float *xrow = x + (size_t)tc * K + 32 * h;

    v16f acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.0f;
    // 64-k chunks, one chunk of loads ahead of the arithmetic (32 MFMAs per block = 2048 issue cycles: an HBM round trip)
    uint4 wq[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) wq[b] = *reinterpret_cast<const uint4 *>(wrow[b] + (k_lo >> 1));
    v4f xv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const v4f *>(xrow + k_lo + 4 * i);
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        uint4 wc[NB];
        v4f xc[8];
#pragma unroll
        for (int b = 0; b < NB; ++b) wc[b] = wq[b];
#pragma unroll
        for (int i = 0; i < 8; ++i) xc[i] = xv[i];
        const int g = (k0 + 32 * h) / group;
        float sc[NB], zp[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) { sc[b] = srow[b][g]; zp[b] = zrow[b][g]; }
        const int kn = (k0 + 64 < k_hi) ? k0 + 64 : k0;             // next chunk (clamped: loads are unconditional)
#pragma unroll
        for (int b = 0; b < NB; ++b) wq[b] = *reinterpret_cast<const uint4 *>(wrow[b] + (kn >> 1));
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[i] = *reinterpret_cast<const v4f *>(xrow + kn + 4 * i);
#pragma unroll
        for (int i = 0; i < 32; ++i) {                              // nibble i of the 16 bytes is k = k0 + 32 h + i
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const uint32_t words[4] = {wc[b].x, wc[b].y, wc[b].z, wc[b].w};
                const float q = (float)((words[i >> 3] >> (4 * (i & 7))) & 0xFu);
                const float a = (q - zp[b]) * sc[b];                // the reference's dq (quantized_linear_kernel.cu:240)
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xc[i >> 2][i & 3], acc[b], 0, 0, 0);
            }
        }
    }
    if constexpr (KSPLIT) {                                         // add the four K quarters (float32: a fixed order)
        __shared__ float red[3][16][64];
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[0][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][r] = ((acc[0][r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
    }
    // D[i][j]: i = weight row (A), j = activation row (B): lane owns t, registers 4q..4q+3 are 4 consecutive n
    if (t >= row_hi) return;
    const bool vec = ((N & 3) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int nq = n_wave + 64 * b + 4 * h + 8 * qd;
            float o[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                o[c] = acc[b][4 * qd + c];
                if (bias != nullptr && nq + c < N) o[c] += bias[(size_t)e * N + nq + c];
            }
            store_out4(out, 0, (size_t)t * N, nq, N, vec, o);
        }
#endif
}
