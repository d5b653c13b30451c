#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof_moe.sh <tag> [bench args...]
# kernel-trace + stats pass, then separate PMC passes (never combined with other trace domains).
tag=$1; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline "$@" > $out/bench_trace.log 2>&1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_INSTS_VALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline "$@" > $out/bench_pmc_$name.log 2>&1 || echo "pmc set failed: $set"
done
python3 tools/summarize_prof.py $out > $out/summary.txt 2>&1 || true
cat $out/summary.txt
