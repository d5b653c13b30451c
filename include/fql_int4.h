/*
 * fql_int4.h -- C ABI of libfql_int4.so: fused INT4 dequantize-linear and grouped (MoE)
 * per-expert INT4 GEMM for AMD Instinct MI355X (gfx950 / CDNA4).
 *
 * This header is the drop-in boundary.  It replaces, entry point for entry point, the two
 * pybind11/libtorch operators of the reference (paths relative to the reference repository):
 *
 *   fql_linear_fwd_f32   <- fused_quant_linear_cuda.forward(input, packed_weights, scales, zero_points)
 *                           csrc/quantized_linear.cpp:22-28, csrc/quantized_linear.h:29-34,
 *                           host wrapper csrc/quantized_linear_kernel.cu:293-378
 *   fql_moe_fwd_f32      <- moe_int4_cuda.forward(packed_weights, scales, zero_points, inputs,
 *                                                 expert_ids, tokens_per_expert, input_offsets)
 *                           csrc/moe_int4_kernel.cu:93-141, csrc/moe_int4_kernel.h:7-15
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / pybind types.  All data pointers are DEVICE
 *     pointers (hipMalloc'd or a framework's device tensor storage) unless stated otherwise.
 *   - the library owns nothing and allocates nothing: outputs and the scratch workspace are
 *     caller-allocated; `*_workspace_bytes` says how much scratch a call needs.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the
 *     null stream), performs no host synchronisation, keeps no state between calls and is
 *     re-entrant.  It is safe to capture into a hipGraph.  (The tuning hooks of fql_int4_tune.h, which no product
 *     path calls, are the one exception: their setters change process-global dispatch thresholds.)
 *   - return value: FQL_OK (0) or a negative FQL_ERR_* code; fql_error_string() describes it.
 *     Nothing is launched when an error is returned.
 *
 * Weight format (identical to the reference; python/quantize.py:120-122, :172):
 *   packed[n][j] = (q[n][2j+1] << 4) | q[n][2j]      uint8, row-major [N][K/2]
 *   w[n][k]      = (q[n][k] - zero_points[n]) * scales[n]          per-ROW scale / zero-point
 */
#ifndef FQL_INT4_H
#define FQL_INT4_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQL_VERSION 210 /* 0.2.1: fql_moe_gather_scaled_fwd_f32, fql_combine_f32 with NULL weights; the 3-limb workspace grew by one flag word per 4 rows (size it with fql_*_workspace_bytes, as always).  0.2.0: fp8 activations; residual limb set for heavy-tailed rows (two-phase buffers doubled) */

#if defined(__GNUC__)
#define FQL_API __attribute__((visibility("default")))
#else
#define FQL_API
#endif

/* error codes */
#define FQL_OK 0
#define FQL_ERR_NULL_POINTER (-1)   /* a required pointer is NULL                                  */
#define FQL_ERR_BAD_SHAPE (-2)      /* negative / zero dimension or size overflow                  */
#define FQL_ERR_ODD_K (-3)          /* K must be even (two weights per byte)                       */
#define FQL_ERR_WORKSPACE (-4)      /* workspace NULL, misaligned (16 B) or smaller than required  */
#define FQL_ERR_LAUNCH (-5)         /* hipGetLastError() reported a launch failure                 */
#define FQL_ERR_BAD_PRECISION (-6)  /* precision not one of FQL_PRECISION_*                        */
#define FQL_ERR_ALIGNMENT (-7)      /* a tensor base pointer is not aligned as documented          */
#define FQL_ERR_DTYPE (-8)          /* element type not supported on the path this shape takes     */

/* Activation precision of the MFMA path.  Weights are always exact (4-bit integers).
 * Activations are split per row into signed 8-bit limbs of a fixed-point value; the integer
 * dot products are exact (i32 MFMA accumulation), so the only error is the one rounding of
 * each activation to 2^-(8*limbs-1) of its row's maximum magnitude:
 *   FQL_PRECISION_EXACT : 3 limbs, 23-bit fixed point -> float32-class results (default;
 *                         meets the reference's own allclose(atol=1e-3) GPU tests)
 *   FQL_PRECISION_FAST  : 2 limbs, 15-bit fixed point -> ~3e-5 relative (Frobenius) error,
 *                         2/3 of the matrix-core work
 *   FQL_PRECISION_INT8  : 1 limb, 8-bit activations (per-row power-of-two scale) -> ~5e-3 relative
 *                         error, 1/3 of the matrix-core work: the "8-bit activations + INT4 weights"
 *                         serving mode (outside the 1e-3 parity claim; for weight-streaming-bound
 *                         shapes such as 64 experts x 7168 -> 18432)                          */
#define FQL_PRECISION_DEFAULT 0
#define FQL_PRECISION_INT8 1
#define FQL_PRECISION_FAST 2
#define FQL_PRECISION_EXACT 3
/*   FQL_PRECISION_FP8   : activations rounded to OCP e4m3 (4-bit significand) with one float32 scale per row
 *                         (max|x| / 448), ONE pass of the block-scaled fp8 matrix-core instruction
 *                         (v_mfma_scale_f32_32x32x64_f8f6f4, float32 accumulation) -- the "fp8 activations + INT4
 *                         weights" configuration of BASELINE.json configs[4].  ~2.7e-2 relative error on randn
 *                         activations (the format's, not the kernel's): outside the 1e-3 parity claim.  MFMA path
 *                         only: K % 32 == 0, 16-byte aligned weights and, for the linear op, more than 2 rows --
 *                         otherwise FQL_ERR_BAD_PRECISION (never a silent float32 computation); a NULL / short
 *                         workspace is FQL_ERR_WORKSPACE. */
#define FQL_PRECISION_FP8 8

/* Element types of activations and outputs for the dtype-generic entry points (fql_linear_fwd / fql_moe_fwd).
 * 16-bit inputs are widened exactly; outputs are rounded to nearest even from the float32 result, i.e. the
 * result equals `fql_*_fwd_f32(x.float()).to(dtype)` bit for bit, without the two conversion passes
 * (reference: benchmark/moe_grouped_gemm/moe_int4_module.py:70-72 runs its experts in the dtype of x). */
#define FQL_DTYPE_F32 0
#define FQL_DTYPE_F16 1
#define FQL_DTYPE_BF16 2

FQL_API int fql_version(void);
FQL_API const char *fql_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * Fused INT4 dequantize-linear:  out[b][n] = sum_k x[b][k] * (q[n][k] - zp[n]) * scale[n]
 *
 *   x       [B][K] float32, contiguous          (reference: `input`, 1-D inputs are B = 1)
 *   packed  [N][K/2] uint8, contiguous
 *   scales  [N] float32,  zps [N] float32
 *   out     [B][N] float32, contiguous, fully overwritten
 *   workspace: fql_linear_workspace_bytes(B, K, N, precision) bytes, 16-byte aligned
 *              (0 bytes for B <= 2: those shapes run on the float32 GEMV kernel and never touch a workspace -- except
 *              with FQL_PRECISION_FP8, where the size is what fql_linear_fwd_f8 needs, which takes any B on the
 *              matrix cores; B = 3, 4 also run with NULL, on the slower GEMV kernel)
 * Any even K is accepted; K % 32 == 0 with 16-byte aligned `packed` takes the fast paths.
 * ------------------------------------------------------------------------------------- */
FQL_API size_t fql_linear_workspace_bytes(int B, int K, int N, int precision);

FQL_API int fql_linear_fwd_f32(const float *x, const uint8_t *packed, const float *scales,
                       const float *zps, float *out, int B, int K, int N, int precision,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Same with a per-column bias added to the result (the reference asserts `bias is None`,
 * python/module.py:84):  out[b][n] = (sum_k x[b][k] * (q[n][k] - zp[n]) * scale[n]) + bias[n],  bias [N] float32 or
 * NULL.  The add is the last operation of the kernels' epilogues (one float32 rounding after the un-biased result). */
FQL_API int fql_linear_bias_fwd_f32(const float *x, const uint8_t *packed, const float *scales,
                                    const float *zps, const float *bias, float *out, int B, int K, int N,
                                    int precision, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Grouped (MoE) INT4 GEMM over rows pre-grouped by expert:
 *   for every expert e:  out[off_e : off_e+cnt_e] = inputs[off_e : off_e+cnt_e] @ dequant(W_e)^T
 *   rows of `out` covered by no expert are set to zero (reference: torch::zeros, :109)
 *
 *   packed  [E][N][K/2] uint8;  scales, zps [E][N] float32
 *   inputs  [T][K] float32;     out [T][N] float32, fully overwritten
 *   tokens_per_expert [E] int32 (cnt_e), input_offsets [E] int32 (off_e): DEVICE arrays,
 *     consumed on the device -- no host read-back, one launch sequence for all experts.
 *     Ranges are clipped to [0, T]; they must not overlap.
 *   (`expert_ids` of the reference signature is accepted by the Python shim and ignored,
 *    exactly as the reference ignores it: csrc/moe_int4_kernel.cu:98.)
 * ------------------------------------------------------------------------------------- */
FQL_API size_t fql_moe_workspace_bytes(int E, int T, int K, int N, int precision);

FQL_API int fql_moe_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                    const float *inputs, const int32_t *tokens_per_expert,
                    const int32_t *input_offsets, float *out, int E, int T, int K, int N,
                    int precision, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Same, with the dispatch gather fused into the activation pre-pass:
 * grouped row t is read from tokens[row_index[t]] -- the sort-by-expert permutation of
 * benchmark/moe_grouped_gemm/routing.py:117-149 -- so the [T, K] gathered copy of the activations is
 * never written or re-read.  tokens [n_tokens][K] float32, row_index [T] int32 (device), values
 * clamped into [0, n_tokens).  Requires the MFMA path (K % 32 == 0, 16-byte aligned `packed`),
 * otherwise FQL_ERR_ALIGNMENT.  Workspace: fql_moe_workspace_bytes(E, T, K, N, precision).
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_moe_gather_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                                   const float *tokens, const int32_t *row_index, int n_tokens,
                                   const int32_t *tokens_per_expert, const int32_t *input_offsets,
                                   float *out, int E, int T, int K, int N, int precision,
                                   void *workspace, size_t workspace_bytes, void *stream);

/* The same with the routing weight folded into the GEMM epilogue (second half;
 * benchmark/moe_grouped_gemm/routing.py:172-189 multiplies after the fact): out[t][:] = row_weight[t] * (grouped result).
 * row_weight [T] float32, one per GROUPED row (the weight of the (token, slot) pair the row belongs to).  One float32
 * rounding after the un-weighted result, so fql_combine_f32(..., weights = NULL), a pure gather-add of these rows, gives
 * bit for bit what fql_combine_f32 with the weights gives on the un-weighted rows (top_k <= 2).  MFMA path only. */
FQL_API int fql_moe_gather_scaled_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                                  const float *tokens, const int32_t *row_index, int n_tokens, const float *row_weight,
                                  const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out,
                                  int E, int T, int K, int N, int precision, void *workspace, size_t workspace_bytes,
                                  void *stream);

/* ---------------------------------------------------------------------------------------
 * Activations that are ALREADY fp8 (OCP e4m3fn bytes, e.g. the output of an upstream fp8 kernel or of
 * torch's .to(torch.float8_e4m3fn)), with an optional float32 scale per row:
 *
 *   out[t][n] = act_scales[t] * scale[e][n] * sum_k (q[e][n][k] - zp[e][n]) * e4m3(inputs[t][k])
 *
 *   inputs_e4m3 [T][K] uint8;  act_scales [T] float32 or NULL (= 1);  out [T][N] of out_dtype (FQL_DTYPE_*)
 * One pass of the block-scaled fp8 matrix-core instruction, float32 accumulation; the 4-bit weights are exact in
 * e4m3.  An e4m3 NaN makes its whole output row NaN.  Not in the reference (FP8 is listed as future work,
 * README.md:228); the shape is BASELINE.json configs[4].  MFMA path only: K % 32 == 0 and 16-byte aligned
 * `packed`, otherwise FQL_ERR_ALIGNMENT.  Workspace: fql_moe_workspace_bytes(E, T, K, N, FQL_PRECISION_FP8)
 * (fql_linear_workspace_bytes(B, ...) for the linear form, which takes any B >= 1).
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_moe_fwd_f8(const uint8_t *packed, const float *scales, const float *zps,
                           const uint8_t *inputs_e4m3, const float *act_scales,
                           const int32_t *tokens_per_expert, const int32_t *input_offsets, void *out,
                           int out_dtype, int E, int T, int K, int N, void *workspace,
                           size_t workspace_bytes, void *stream);

FQL_API int fql_linear_fwd_f8(const uint8_t *x_e4m3, const float *act_scales, const uint8_t *packed,
                              const float *scales, const float *zps, void *out, int out_dtype, int B, int K,
                              int N, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Per-GROUP scales and zero points along K (the layout GPTQ / AWQ-style INT4 checkpoints use;
 * NOT in the reference, whose quantisation is per output row, python/quantize.py:73-80):
 *
 *   out[t][n] = sum_k x[t][k] * (q[n][k] - zps[n][k / group_size]) * scales[n][k / group_size]   (+ bias[n])
 *
 *   scales, zps [N][K / group_size] float32 ([E][N][K / group_size] for the grouped form); group_size even, divides K.
 * Any shape, no workspace.  The weights are dequantised in registers, (q - zp) * scale as the reference kernel does per
 * element, and the contraction is float32: up to 3 rows of ONE matrix (K % 32 == 0, group_size % 32 == 0, 16-byte
 * aligned bases) on the GEMV kernel with the constants folded per 32-k chunk; 4 or more rows per group (K % 64 == 0,
 * group_size % 32 == 0, aligned bases) on the float32 matrix-core instruction; one wave per output row (float32 FMA)
 * otherwise.  The integer MFMA kernels need a single scale per output row: per-row quantisation (group_size == K) stays
 * on the faster entry points above.
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_linear_group_fwd_f32(const float *x, const uint8_t *packed, const float *scales,
                                     const float *zps, const float *bias, float *out, int B, int K, int N,
                                     int group_size, void *stream);

FQL_API int fql_moe_group_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                                  const float *inputs, const int32_t *tokens_per_expert,
                                  const int32_t *input_offsets, float *out, int E, int T, int K, int N,
                                  int group_size, void *stream);

/* The same two calls WITH a workspace (fql_group_workspace_bytes bytes, 16-byte aligned): batches (40 or more rows of one
 * matrix, 8 or more rows per expert) with K % 256 == 0 and group_size % 64 == 0 then run on the INT8 matrix cores -- the activation limbs of the
 * per-row path (`precision` as there: exact / fast / int8), integer dot products and limb sums per group, folded in
 * float32 with the group's scale and zero point at the end of every group (csrc/fql_group_i8.h); 2.5-3x the per-row
 * path's time instead of 6x.  Every other shape, or a NULL / short workspace, takes the float32 paths above. */
FQL_API size_t fql_group_workspace_bytes(int E, int T, int K, int N, int group_size, int precision);

FQL_API int fql_linear_group_ws_fwd_f32(const float *x, const uint8_t *packed, const float *scales,
                                        const float *zps, const float *bias, float *out, int B, int K, int N,
                                        int group_size, int precision, void *workspace, size_t workspace_bytes,
                                        void *stream);

FQL_API int fql_moe_group_ws_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                                     const float *inputs, const int32_t *tokens_per_expert,
                                     const int32_t *input_offsets, float *out, int E, int T, int K, int N,
                                     int group_size, int precision, void *workspace, size_t workspace_bytes,
                                     void *stream);

/* ---------------------------------------------------------------------------------------
 * Format helpers on the device (same unpack code path as the GEMM kernels; bit-exact).
 *   fql_unpack_u8     : q[i][2j] = packed[i][j] & 15, q[i][2j+1] = packed[i][j] >> 4
 *                       (python/quantize.py:152-163)
 *   fql_dequantize_f32: w[n][k] = (q[n][k] - zps[n]) * scales[n]   (python/quantize.py:172)
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_unpack_u8(const uint8_t *packed, uint8_t *q, size_t nbytes, void *stream);

FQL_API int fql_dequantize_f32(const uint8_t *packed, const float *scales, const float *zps, float *w,
                       int N, int K, void *stream);

/* ---------------------------------------------------------------------------------------
 * Quantisers on the device, bit-exact with the reference's host arithmetic:
 *   fql_quantize_rows_f32   : python/quantize.py:38-124  per-ROW scale / zero-point (constant-row guard,
 *                             1e-8 floor, round-half-to-even), packed [N][K/2], scales / zps [N]
 *   fql_quantize_tensor_f32 : python/moe_int4_module.py:45-76  per-TENSOR scale / zero-point broadcast to
 *                             [N] (one expert); scratch = 2*N floats of device memory
 *   w [N][K] float32 contiguous (fp16 weights are widened by the caller, as the reference does).
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_quantize_rows_f32(const float *w, uint8_t *packed, float *scales, float *zps, int N, int K,
                                  void *stream);

FQL_API int fql_quantize_tensor_f32(const float *w, uint8_t *packed, float *scales, float *zps,
                                    float *scratch, int N, int K, void *stream);

/* ---------------------------------------------------------------------------------------
 * Phase 1 of the MFMA path on its own: the activation pre-pass (also the tests' window into it).
 *   x[t][k] ~= delta[t] * sum_l 256^l * a_l[t][k],  a_l signed 8-bit ("limbs"), delta a power of two
 *   limbs   fql_act_limb_bytes(T, E, K, precision) bytes, 16-byte aligned, in MFMA-fragment order:
 *           [limb][k / 256][padded_row / 32][k-step 0..7][lane 0..63][16 B]  (see csrc/fql_act_quant.h;
 *           expert e's rows start at padded row sum_{e'<e} roundup(cnt_e', 32); K is zero-padded to
 *           fql_act_padded_k(K))
 *   delta   [T] float32, rowsum [limbs][T] int32 (sum over k of each limb)
 *   tokens_per_expert / input_offsets: device arrays as in fql_moe_fwd_f32, or both NULL with E = 1
 *   for one group covering all T rows.  Rows covered by no expert are skipped.
 *
 * Heavy-tailed rows (FQL_PRECISION_FAST / _EXACT / _DEFAULT): the limbs are per-ROW block fixed point, so one
 * outlier element coarsens the quantum of its whole row.  A row whose 8*limbs-1 bits would not carry the mode's
 * stated precision -- predicted relative output error sqrt(K/12) / ||x/delta||_2 above 1e-6 (3 limbs) or
 * 2.5e-4 (2 limbs); randn rows never are -- gets a SECOND limb set holding the rounding residual
 * (x/delta - X) * 2^(8*limbs-1), quantum delta2 = delta * 2^-(8*limbs-1), which the GEMM adds for that row
 * (16*limbs-2 bits in all).  For these precisions the buffers are therefore doubled:
 *   limbs   fql_act_limb_bytes() bytes = both sets back to back;  delta [2][T] (delta, delta2; delta2 = 0
 *   marks a row without residual);  rowsum [2][limbs][T].
 * FQL_PRECISION_INT8 / _FP8 keep one set.  With _FP8 the plane holds e4m3 bytes, delta the per-row scale
 * max|x| / 448 and rowsum[0] the float32 bits of the row sum.
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_act_padded_k(int K);

FQL_API size_t fql_act_limb_bytes(int T, int E, int K, int precision);

FQL_API int fql_act_quant_f32(const float *x, int8_t *limbs, float *delta, int32_t *rowsum,
                              const int32_t *tokens_per_expert, const int32_t *input_offsets, int E,
                              int T, int K, int precision, void *stream);

/* ---------------------------------------------------------------------------------------
 * Phase 2 of the MFMA path on its own: the grouped INT4 x INT8-limb GEMM over activations that
 * fql_act_quant_f32 already converted (fql_linear_fwd_f32 / fql_moe_fwd_f32 = phase 1 + phase 2
 * back to back).  Lets a caller fuse its own gather/dispatch into phase 1, and lets bench.py
 * time the dominant kernel alone.  The expert arrays must be the ones phase 1 was given (they fix
 * the limb layout); tokens_per_expert == NULL means one group covering all T rows (E must be 1).  Requires K % 32 == 0 and a 16-byte aligned `packed`
 * (FQL_ERR_ALIGNMENT otherwise).  Rows covered by no expert are left untouched.
 * `scratch`: fql_gemm_scratch_bytes(precision) bytes, 16-byte aligned -- workgroup-private float32 partials of the
 * residual pass over tiles that hold heavy-tailed rows (see phase 1).  With scratch == NULL (or too small) that
 * pass is skipped and such rows keep the plain 8*limbs-1 bit result.
 * ------------------------------------------------------------------------------------- */
FQL_API size_t fql_gemm_scratch_bytes(int precision);

FQL_API int fql_gemm_i8_f32(const int8_t *limbs, const float *delta, const int32_t *rowsum,
                            const uint8_t *packed, const float *scales, const float *zps,
                            const int32_t *tokens_per_expert, const int32_t *input_offsets,
                            float *out, int E, int T, int K, int N, int precision, void *stream,
                            void *scratch, size_t scratch_bytes);

/* ---------------------------------------------------------------------------------------
 * Dtype-generic forms of fql_linear_fwd_f32 / fql_moe_fwd_f32 (float16 / bfloat16
 * activations and outputs, as the reference's MoE benches feed them).  Same arguments plus the element
 * types; (F32, F32) forwards to the float32 entry points.  16-bit I/O exists on the MFMA path only
 * (more than 2 rows, K % 32 == 0, 16-byte aligned packed weights): other shapes return FQL_ERR_DTYPE and
 * the caller converts.  fql_native_dtype_supported answers that question without launching anything.
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_native_dtype_supported(int rows, int E, int K, int N, int precision, const void *packed,
                                       int grouped);

FQL_API int fql_linear_fwd(const void *x, int in_dtype, const uint8_t *packed, const float *scales,
                           const float *zps, void *out, int out_dtype, int B, int K, int N, int precision,
                           void *workspace, size_t workspace_bytes, void *stream);

FQL_API int fql_moe_fwd(const uint8_t *packed, const float *scales, const float *zps, const void *inputs,
                        int in_dtype, const int32_t *tokens_per_expert, const int32_t *input_offsets,
                        void *out, int out_dtype, int E, int T, int K, int N, int precision,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Second GEMM of a gated FFN expert with the activation fused in (the reference models
 * only the up projection, benchmark/moe_grouped_gemm/config.py:50-52):
 *   out[t][:] = W_e * ( silu(gate_up[t][0:K]) (.) gate_up[t][K:2K] )
 * gate_up is the [T, 2K] float32 output of the fused gate|up projection (one grouped GEMM over the stacked
 * [E, 2K, H] weights); the [T, K] hidden activation is never written.  tokens_per_expert / input_offsets as in
 * fql_moe_fwd_f32, or both NULL with E = 1 for a dense layer.  MFMA path only (K % 32 == 0, 16-byte aligned
 * packed weights): FQL_ERR_ALIGNMENT otherwise.  Workspace: fql_moe_workspace_bytes(E, T, K, N, precision).
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_moe_gated_fwd_f32(const uint8_t *packed, const float *scales, const float *zps,
                                  const float *gate_up, const int32_t *tokens_per_expert,
                                  const int32_t *input_offsets, float *out, int E, int T, int K, int N,
                                  int precision, void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Routing either side of the grouped GEMM, one launch each (reference: the torch index ops of
 * benchmark/moe_grouped_gemm/routing.py:117-149 create_expert_inputs and :172-189 combine_expert_outputs).
 *
 * fql_route_plan_i32: stable counting sort of the n_slots = tokens * top_k (token, slot) pairs by expert id
 *   (expert_of_slot = expert_indices flattened, ids clamped into [0, E), E <= 128).  Outputs, all int32 on the
 *   device: counts[E] and offsets[E] (= tokens_per_expert / input_offsets), token_of_sorted[n_slots] (the
 *   row_index of fql_moe_gather_fwd_f32) and pos_of_slot[n_slots] (where each slot's result row lands).
 * fql_combine_f32: out[t][:] = sum_{k < top_k} weights[t][k] * y[pos_of_slot[t*top_k + k]][:], k ascending;
 *   y is [R, N], out [T, N] (T <= 65535).  weights == NULL: a pure gather-add of rows that already carry their
 *   weight (fql_moe_gather_scaled_fwd_f32).
 * fql_regroup_index_i32 (expert-parallel receive side): recv_counts[G][EL] rows per (source rank, local expert)
 *   in arrival order -> tokens_per_expert[EL], input_offsets[EL], gather[R] (expert-major position -> received
 *   row, the row_index of fql_moe_gather_fwd_f32) and scatter[R] (its inverse); G * EL <= 8192.
 * ------------------------------------------------------------------------------------- */
FQL_API int fql_route_plan_i32(const int32_t *expert_of_slot, int n_slots, int top_k, int E, int32_t *counts,
                               int32_t *offsets, int32_t *token_of_sorted, int32_t *pos_of_slot,
                               void *stream);

FQL_API int fql_combine_f32(const float *y, const int32_t *pos_of_slot, const float *weights, float *out,
                            int T, int top_k, int N, int R, void *stream);

FQL_API int fql_regroup_index_i32(const int32_t *recv_counts, int G, int EL, int32_t *tokens_per_expert,
                                  int32_t *input_offsets, int32_t *gather, int32_t *scatter, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FQL_INT4_H */
