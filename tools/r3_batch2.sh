#!/bin/bash
# Usage: tools/r3_batch2.sh <outdir>   (one gpurun call: GPU test suite, bench, traces, A/B, LDS counters)
out=$1; mkdir -p $out
export TMPDIR=/tmp
set -x
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1 || { tail -30 $out/pytest_gpu.log; exit 1; }
tail -3 $out/pytest_gpu.log
timeout -k 10 300 python bench.py > $out/bench_default.json 2> $out/bench_default.err || { tail -20 $out/bench_default.err; exit 1; }
timeout -k 10 300 python bench.py --routing skewed --no-cpu-baseline > $out/bench_skewed.json 2> $out/bench_skewed.err || exit 1
timeout -k 10 300 python bench.py --workload linear512 --no-cpu-baseline > $out/bench_linear512.json 2> $out/bench_linear512.err || exit 1
export FQL_INT4_LIB=tools/micro/libfql_trace.so
for cfg in 300 301; do
  timeout -k 10 100 python tools/trace_w4.py --cfg $cfg > $out/trace_${cfg}.log 2>&1 || exit 1
done
unset FQL_INT4_LIB
timeout -k 10 200 python tools/tune_gemm.py --cfgs 0,300,301 --rounds 9 > $out/tune_balanced.log 2>&1 || exit 1
timeout -k 10 200 python tools/tune_gemm.py --cfgs 0,300,301 --rounds 5 --routing skewed > $out/tune_skewed.log 2>&1 || exit 1
timeout -k 10 200 python tools/tune_gemm.py --cfgs 0,300,301 --rounds 5 --workload linear512 > $out/tune_linear512.log 2>&1 || exit 1
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 tools/tune_gemm.py --cfgs 0,300,301 --rounds 1 --iters 4 > $out/pmc_$name.log 2>&1 || echo "pmc set failed: $set"
done
python3 tools/summarize_prof.py $out > $out/pmc_summary.txt 2>&1 || true
echo done
