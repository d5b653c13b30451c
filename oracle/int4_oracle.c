/*
 * CPU ORACLE (plain C) -- TEST INFRASTRUCTURE ONLY.  Not product code.
 *
 * Scalar C restatement of the reference's fused INT4 dequantize-linear / MoE expert GEMM
 * hot path.  Built by oracle/Makefile into oracle/libint4_oracle.so and loaded with ctypes
 * by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg ONLY.  The shipped
 * library (csrc/libfql_int4.so) neither links nor calls anything in this file.
 *
 * Parity status: PINNED against the reference's own Python run in the build container
 * (the .npz vectors under tests/golden/, see tests/test_oracle_golden.py).
 *
 * Reference lines restated (paths relative to the reference repository):
 *   python/quantize.py:38-124            quantize_weights        -> oracle_quantize_rows
 *   python/moe_int4_module.py:19-80      quantize_weights_moe    -> oracle_quantize_tensor
 *   python/quantize.py:127-173           dequantize_weights      -> oracle_unpack / oracle_dequantize
 *   python/quantize.py:176-202           reference_quantized_linear -> oracle_linear_f64acc
 *   csrc/quantized_linear_kernel.cu:218-264  the CUDA kernel's own fmaf chain -> oracle_linear_fma
 *   csrc/moe_int4_kernel.cu:93-136 (intended semantics), python/moe_int4_module.py:122-146
 *                                                                -> oracle_moe_grouped
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* torch.round / np.rint: round half to even.  nearbyintf under the default FE_TONEAREST mode. */
static inline float rne(float v) { return nearbyintf(v); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* python/quantize.py:38-124.  weight [N,K] f32 -> packed [N,K/2] u8, scales [N], zps [N]. */
API int oracle_quantize_rows(const float *w, int N, int K, uint8_t *packed, float *scales, float *zps)
{
    if (N < 0 || K < 0 || (K & 1)) return -1;                 /* :63-64 */
    for (int n = 0; n < N; ++n) {
        const float *row = w + (size_t)n * K;
        float mn = row[0], mx = row[0];                       /* :73-74 */
        for (int k = 1; k < K; ++k) { if (row[k] < mn) mn = row[k]; if (row[k] > mx) mx = row[k]; }
        float scale = (mx - mn) / 15.0f;                      /* :80 */
        if (mx == mn) {                                       /* :85-92 */
            float a = fabsf(mx);
            scale = (a < 1.0f ? 1.0f : a) / 15.0f;
        }
        if (scale < 1e-8f) scale = 1e-8f;                     /* :94 */
        float zp = clampf(rne(-mn / scale), 0.0f, 15.0f);     /* :100-101 */
        scales[n] = scale;
        zps[n] = zp;
        uint8_t *prow = packed + (size_t)n * (K / 2);
        for (int j = 0; j < K / 2; ++j) {                     /* :106-109, :120-122 */
            float q0 = clampf(rne(row[2 * j] / scale + zp), 0.0f, 15.0f);
            float q1 = clampf(rne(row[2 * j + 1] / scale + zp), 0.0f, 15.0f);
            prow[j] = (uint8_t)(((uint8_t)q1 << 4) | (uint8_t)q0);
        }
    }
    return 0;
}

/* python/moe_int4_module.py:45-76, one expert.  w is float32 (the caller widened fp16). */
API int oracle_quantize_tensor(const float *w, int N, int K, uint8_t *packed, float *scale_out, float *zp_out)
{
    if (N <= 0 || K <= 0 || (K & 1)) return -1;
    size_t total = (size_t)N * K;
    float mn = w[0], mx = w[0];
    for (size_t i = 1; i < total; ++i) { if (w[i] < mn) mn = w[i]; if (w[i] > mx) mx = w[i]; }
    float scale = (mx - mn) / 15.0f;                          /* :49 */
    double zq = nearbyint((double)(-mn / scale));             /* :50 python round(), half-to-even */
    float zp = (float)(zq < 0.0 ? 0.0 : (zq > 15.0 ? 15.0 : zq));   /* :51 */
    *scale_out = scale;
    *zp_out = zp;
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < K / 2; ++j) {                     /* :57-59, :62-76 */
            const float *row = w + (size_t)n * K;
            float q0 = clampf(rne(row[2 * j] / scale + zp), 0.0f, 15.0f);
            float q1 = clampf(rne(row[2 * j + 1] / scale + zp), 0.0f, 15.0f);
            packed[(size_t)n * (K / 2) + j] = (uint8_t)(((uint8_t)q1 << 4) | (uint8_t)q0);
        }
    return 0;
}

/* python/quantize.py:152-163: q[2j] = byte & 0xF, q[2j+1] = byte >> 4. */
API void oracle_unpack(const uint8_t *packed, size_t nbytes, uint8_t *q)
{
    for (size_t j = 0; j < nbytes; ++j) { q[2 * j] = packed[j] & 0x0F; q[2 * j + 1] = packed[j] >> 4; }
}

/* python/quantize.py:172: (q - zp) * scale in float32. */
API void oracle_dequantize(const uint8_t *packed, const float *scales, const float *zps,
                           int N, int K, float *w)
{
    for (int n = 0; n < N; ++n)
        for (int j = 0; j < K / 2; ++j) {
            uint8_t b = packed[(size_t)n * (K / 2) + j];
            w[(size_t)n * K + 2 * j]     = ((float)(b & 0x0F) - zps[n]) * scales[n];
            w[(size_t)n * K + 2 * j + 1] = ((float)(b >> 4) - zps[n]) * scales[n];
        }
}

/* python/quantize.py:201-202 with the float32 dequantised weights accumulated in float64:
 * the order-independent ground truth both the sgemm oracle and the GPU kernel are judged by. */
API void oracle_linear_f64acc(const float *x, const uint8_t *packed, const float *scales,
                              const float *zps, int B, int K, int N, double *out)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const uint8_t *prow = packed + (size_t)n * (K / 2);
            const float *xr = x + (size_t)b * K;
            double acc = 0.0;
            for (int j = 0; j < K / 2; ++j) {
                float w0 = ((float)(prow[j] & 0x0F) - zps[n]) * scales[n];
                float w1 = ((float)(prow[j] >> 4) - zps[n]) * scales[n];
                acc += (double)w0 * (double)xr[2 * j];
                acc += (double)w1 * (double)xr[2 * j + 1];
            }
            out[(size_t)b * N + n] = acc;
        }
}

/* csrc/quantized_linear_kernel.cu:240-244: dq = fmaf(q - zp, scale, 0); sum = fmaf(dq, x, sum),
 * sequential in k, float32 -- what the reference's own GPU kernel computes. */
API void oracle_linear_fma(const float *x, const uint8_t *packed, const float *scales,
                           const float *zps, int B, int K, int N, float *out)
{
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < N; ++n) {
            const uint8_t *prow = packed + (size_t)n * (K / 2);
            const float *xr = x + (size_t)b * K;
            float sum = 0.0f;
            for (int j = 0; j < K / 2; ++j) {
                float dq0 = fmaf((float)(prow[j] & 0x0F) - zps[n], scales[n], 0.0f);
                float dq1 = fmaf((float)(prow[j] >> 4) - zps[n], scales[n], 0.0f);
                sum = fmaf(dq0, xr[2 * j], sum);
                sum = fmaf(dq1, xr[2 * j + 1], sum);
            }
            out[(size_t)b * N + n] = sum;
        }
}

/* Intended contract of csrc/moe_int4_kernel.cu:93-136 / python/moe_int4_module.py:122-146:
 * rows pre-grouped by expert; out[off_e:off_e+cnt_e] = in[...] @ dequant(W_e)^T; rest zero
 * (torch::zeros, :109); expert_ids ignored (:98).  float64 accumulation. */
API int oracle_moe_grouped(const uint8_t *packed, const float *scales, const float *zps,
                           const float *inputs, const int32_t *tokens_per_expert,
                           const int32_t *input_offsets, int E, int T, int K, int N, double *out)
{
    memset(out, 0, sizeof(double) * (size_t)T * N);
    for (int e = 0; e < E; ++e) {
        int cnt = tokens_per_expert[e], off = input_offsets[e];
        if (cnt <= 0) continue;
        if (off < 0 || off + cnt > T) return -2;
        oracle_linear_f64acc(inputs + (size_t)off * K, packed + (size_t)e * N * (K / 2),
                             scales + (size_t)e * N, zps + (size_t)e * N, cnt, K, N,
                             out + (size_t)off * N);
    }
    return 0;
}
