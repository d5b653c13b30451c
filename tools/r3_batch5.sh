#!/bin/bash
# round-3 batch 5 (GPU box): one-launch form -- tests, A/B bench, timeline
o=gpurun_out/r3aa; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_w4.py -x -q -m gpu > $o/pytest.log 2>&1 || { tail -40 $o/pytest.log; exit 1; }
tail -3 $o/pytest.log
for i in 1 2; do
  for m in 0 1; do
    timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-side-modes --no-cpu-baseline --one-launch $m > $o/bench_${m}_$i.json 2>$o/err.log || { tail -20 $o/err.log; exit 1; }
    python -c "
import json; d=json.load(open('$o/bench_${m}_$i.json')); print('one-launch $m run $i: step %.1f us' % (d['ms_per_step']*1e3), 'max_rel_err', d.get('max_rel_err'))"
  done
done
FQL_INT4_LIB=tools/micro/libfql_trace.so timeout -k 10 120 python tools/trace_step.py --one-launch > $o/trace_step_fused.log 2>&1 || { tail -20 $o/trace_step_fused.log; exit 1; }
sed -n 2,12p $o/trace_step_fused.log | cut -c1-200
grep "gemm wg [0-3]" $o/trace_step_fused.log | cut -c1-220
