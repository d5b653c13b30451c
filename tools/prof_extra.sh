#!/bin/bash
# rocprofv3 kernel-trace + stats and a FETCH/WRITE PMC pass for the fp8 configs[4] workload and the decode workload
set -o pipefail
export TMPDIR=/tmp
for spec in "config5_fp8:--experts 64 --hidden 7168 --ffn 18432 --tokens 512 --top-k 6 --weight-sets 2 --precision fp8" "decode32:--tokens 32"; do
  tag=${spec%%:*}; args=${spec#*:}
  out=gpurun_out/prof_${PROF_ROUND:-r03}_$tag; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side-modes $args > $out/bench_trace.log 2>&1
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-side-modes $args > $out/bench_pmc_$name.log 2>&1 || echo "pmc set failed: $set"
  done
  python3 tools/summarize_prof.py $out > $out/summary.txt 2>&1 || true
  echo "== $tag"; cat $out/summary.txt
done
