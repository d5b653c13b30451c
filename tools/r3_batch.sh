#!/bin/bash
# One gpurun call, many measurements (box acquisition is the expensive part).  Usage: tools/r3_batch.sh <outdir>
out=$1; mkdir -p $out
export TMPDIR=/tmp
set -x
timeout -k 10 200 ./tools/micro/mfma_order_probe > $out/mfma_order_probe.log 2>&1 || exit 1
export FQL_INT4_LIB=tools/micro/libfql_trace.so
for cfg in 300 301 0; do
  timeout -k 10 100 python tools/trace_w4.py --cfg $cfg > $out/trace_${cfg}.log 2>&1 || exit 1
  timeout -k 10 100 python tools/trace_w4.py --cfg $cfg --abs-acts > $out/trace_${cfg}_abs.log 2>&1 || exit 1
done
timeout -k 10 100 python tools/trace_w4.py --cfg 300 --zero-acts > $out/trace_300_zero.log 2>&1 || exit 1
unset FQL_INT4_LIB
timeout -k 10 200 python tools/tune_gemm.py --cfgs 0,300,301 --rounds 9 > $out/tune_balanced.log 2>&1 || exit 1
timeout -k 10 200 python tools/tune_gemm.py --cfgs 0,300,301 --rounds 5 --routing skewed > $out/tune_skewed.log 2>&1 || exit 1
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $out/pmc_$name -- python3 tools/tune_gemm.py --cfgs 0,300,301 --rounds 1 --iters 4 > $out/pmc_$name.log 2>&1 || echo "pmc set failed: $set"
done
python3 tools/summarize_prof.py $out > $out/pmc_summary.txt 2>&1 || true
echo done
