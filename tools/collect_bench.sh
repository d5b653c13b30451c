#!/bin/bash
# Bench lines for profiles/ (run on the GPU box from the repo root): bash tools/collect_bench.sh [tag]   (default tag r03)
tag=${1:-r03}; o=gpurun_out/bench_$tag; mkdir -p $o
run() { name=$1; shift; timeout -k 10 400 python bench.py "$@" 2>/dev/null | grep '^{' > $o/$name.json; python - <<PY
import json
try:
    d = json.load(open("$o/$name.json"))
    r = d.get("roofline") or {}
    print("%-22s step %8.1f us  value %7.1f %s  roofline %s %.3f  gemm %.1f us  max_rel_err %.2e" % ("$name", d["ms_per_step"]*1e3, d["value"], d["unit"], r.get("bound"), r.get("frac", 0), d.get("gemm_kernel_ms_avg", 0)*1e3, d.get("max_rel_err", float("nan"))))
except Exception as e:
    print("$name FAILED", e)
PY
}
run moe --steps 100 --warmup 20
run moe_fast --steps 100 --warmup 20 --precision fast --no-cpu-baseline --no-side-modes
run moe_int8 --steps 100 --warmup 20 --precision int8 --no-cpu-baseline --no-side-modes
run moe_fp8 --steps 100 --warmup 20 --precision fp8 --no-cpu-baseline --no-side-modes
run moe_skewed --steps 100 --warmup 20 --routing skewed --no-cpu-baseline --no-side-modes
run moe_decode32 --steps 200 --warmup 20 --tokens 32 --no-cpu-baseline --no-side-modes
run moe_top1 --steps 100 --warmup 20 --top-k 1 --no-cpu-baseline --no-side-modes
run linear512 --steps 100 --warmup 20 --workload linear512
run linear1 --steps 200 --warmup 20 --workload linear1
run harness_a12 --steps 30 --warmup 5 --experts 8 --hidden 4096 --ffn 14336 --tokens 4096 --top-k 2 --no-cpu-baseline --no-side-modes --weight-sets 2
run config5_exact --steps 20 --warmup 5 --experts 64 --hidden 7168 --ffn 18432 --tokens 512 --top-k 6 --weight-sets 2 --no-cpu-baseline --no-side-modes
run config5_int8 --steps 20 --warmup 5 --experts 64 --hidden 7168 --ffn 18432 --tokens 512 --top-k 6 --weight-sets 2 --precision int8 --no-cpu-baseline --no-side-modes
run config5_fp8 --steps 20 --warmup 5 --experts 64 --hidden 7168 --ffn 18432 --tokens 512 --top-k 6 --weight-sets 2 --precision fp8 --no-cpu-baseline --no-side-modes
