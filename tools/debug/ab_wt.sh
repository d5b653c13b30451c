#!/bin/bash
# A/B of the write-through limb stores through bench.py's step time (alternating rounds; compare the minima)
for r in 1 2 3 4; do
for v in base limbwt; do
  if [ $v = base ]; then unset FQL_INT4_LIB; else export FQL_INT4_LIB=tools/micro/libfql_$v.so; fi
  FQL_BENCH_SKIP_CHECK=1 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-side-modes 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$v round $r: step %.1f us  gemm avg %.1f med %.1f us  prepass %.1f us' % (d['ms_per_step']*1e3, d['gemm_kernel_ms_avg']*1e3, d['gemm_kernel_ms_median']*1e3, d['act_quant_ms_avg']*1e3))"
done; done
