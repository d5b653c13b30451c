/* libfql_int4 -- tuning and diagnostic hooks.  NOT part of the drop-in boundary (include/fql_int4.h): nothing in the
 * product path calls them.  They exist for tools/ (A/B timing in one process), tests/ (every tile configuration must
 * return the same bits) and bench.py (labels its roofline with the kernel family the library picked).
 *
 * The setters change PROCESS-GLOBAL dispatch thresholds of the library: they are not thread-safe, and a process that calls
 * one no longer has the "no state between calls" property fql_int4.h promises -- use them from single-threaded tools only,
 * and restore the returned previous value.  The getters and fql_tune_gemm_i8_f32 keep no state.
 */
#ifndef FQL_INT4_TUNE_H
#define FQL_INT4_TUNE_H

#include "fql_int4.h"

#ifdef __cplusplus
extern "C" {
#endif

/* fql_gemm_i8_f32 with an explicit tile configuration id: 0.. wide 8-wave tiles, 100.. 32-row tiles, 200.. / 220.. 16-row
 * tiles, 300.. the one-wave-per-SIMD kernel.  FQL_ERR_BAD_SHAPE for an id that does not exist for the precision. */
FQL_API int fql_tune_gemm_i8_f32(int cfg, const int8_t *limbs, const float *delta, const int32_t *rowsum,
                                 const uint8_t *packed, const float *scales, const float *zps,
                                 const int32_t *tokens_per_expert, const int32_t *input_offsets, float *out, int E,
                                 int T, int K, int N, int precision, void *stream, void *scratch, size_t scratch_bytes);
/* the id the product path picks for a shape (grouped: rows are spread over E experts) */
FQL_API int fql_tune_chosen_cfg(int precision, int E, int T, int K, int N, int grouped);
FQL_API int fql_tune_num_configs(void);                 /* wide ids run 0 .. this - 1; not every id is built for every precision: */
FQL_API int fql_tune_is_config(int cfg, int precision);  /* 1 when fql_tune_gemm_i8_f32 accepts the id for the precision */
FQL_API int fql_tune_num_rows32_configs(void);
FQL_API int fql_tune_num_rows16_configs(void);
FQL_API int fql_tune_num_w4_configs(void);

/* process-global switches; each returns the previous value */
FQL_API int fql_tune_set_compute_units(int n);          /* persistent grids sized for n compute units (multiple of 8; 0: the device's); tests use it to walk many tiles per workgroup on small shapes */
FQL_API int fql_tune_set_fused(int on);                  /* 3 limbs, float32 rows, one-wave-per-SIMD kernel: pre-pass as the GEMM kernel's first phase (ONE launch) */
FQL_API int fql_tune_set_fused_spin(int polls);          /* polls before a workgroup of that launch quantises the rows it waits for itself (0: at once -- tests) */
FQL_API int fql_tune_set_w4(int on);                     /* 3 limbs, > 64 rows per group: one-wave-per-SIMD kernel (default on) */
FQL_API int fql_tune_set_balance_tiles(int on);          /* uneven column tiles that even out the persistent walk (default on) */
FQL_API int fql_tune_set_gemv_max_rows(int rows);        /* linear op: rows up to which the float32 GEMV kernel runs (default 2) */
FQL_API int fql_tune_set_act_single_rows(int rows);      /* pre-pass: one row per workgroup up to this many padded rows (512) */
FQL_API int fql_tune_set_group_mfma(int on);             /* per-group scales: float32 matrix-core kernel for batches */
FQL_API int fql_tune_set_group_i8(int on);               /* per-group scales: INT8 matrix-core kernel where eligible */
FQL_API int fql_tune_set_group_i8_min_rows(int rows);    /* ... from this many rows per group on */

#ifdef __cplusplus
}
#endif
#endif
