#!/bin/bash
# round-3 batch 4 (GPU box): grouped-row pre-pass + prologue changes -- full GPU suite, step trace, bench lines
o=gpurun_out/r3y; mkdir -p $o
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1 || { tail -40 $o/pytest.log; exit 1; }
tail -3 $o/pytest.log
FQL_INT4_LIB=tools/micro/libfql_trace.so timeout -k 10 120 python tools/trace_step.py > $o/trace_step.log 2>&1 || { tail -20 $o/trace_step.log; exit 1; }
timeout -k 10 400 python bench.py --steps 100 --warmup 20 > $o/bench_moe.json 2> $o/bench_moe.err || { tail -20 $o/bench_moe.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3y/bench_moe.json"))
print("moe step %.1f us gemm %.1f us act %.1f us frac %.3f" % (d["ms_per_step"]*1e3, d.get("gemm_kernel_ms_avg",0)*1e3, d.get("act_quant_ms_avg",0)*1e3, d["roofline"]["frac"]))
for k in ("skewed_routing", "decode32", "linear1", "single_assignment_top1"):
    if k in d: print(k, {kk: (round(v, 4) if isinstance(v, float) else v) for kk, v in d[k].items() if kk != "note" and kk != "tokens_per_expert"})
PY
