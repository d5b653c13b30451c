#!/bin/bash
# round-3 evidence (GPU box, repo root): rocprofv3 summaries of the headline and of the decode shape, bench lines, batch sweep
set -o pipefail
bash tools/prof_moe.sh r03 --no-side-modes > gpurun_out/prof_r03.log 2>&1 || { tail -5 gpurun_out/prof_r03.log; exit 1; }
echo "prof_moe done"
PROF_ROUND=r03 bash tools/prof_extra.sh > gpurun_out/prof_r03_extra.log 2>&1 || { tail -5 gpurun_out/prof_r03_extra.log; exit 1; }
echo "prof_extra done"
bash tools/collect_bench.sh r03b > gpurun_out/bench_r03b_summary.txt 2>&1 || { tail -5 gpurun_out/bench_r03b_summary.txt; exit 1; }
cat gpurun_out/bench_r03b_summary.txt
timeout -k 10 300 python tools/time_linear.py > gpurun_out/r03_linear_batch_sweep.txt 2>&1 || { tail -5 gpurun_out/r03_linear_batch_sweep.txt; exit 1; }
tail -14 gpurun_out/r03_linear_batch_sweep.txt
