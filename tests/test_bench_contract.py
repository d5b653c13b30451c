"""bench.py's output contract on a GPU box: one JSON line with the required keys at N = 1, and the multi-rank
code path rehearsed with two ranks sharing the card (collectives through host memory over gloo: a rehearsal of
the control flow the driver will run under torch.distributed.run with RCCL, never a measurement)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]


def _last_json(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line():
    p = subprocess.run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-side-modes", "--tokens", "64"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and 0 < r["frac"] <= 1 and r["unit"] in ("GB/s", "TFLOP/s")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0


def test_two_rank_rehearsal():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, FQL_BENCH_BACKEND="gloo")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2",
                        "--steps", "3", "--warmup", "1", "--weight-sets", "1"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["parallelism"] == "ep2"
    assert set(d["ep_phases_ms_max_over_ranks"]) >= {"dispatch_all_to_all", "regroup_and_grouped_gemm", "combine_all_to_all"}
    assert d["roofline"]["frac"] > 0
    par = d["parity_vs_single_gpu"]                      # every rank's sampled rows == the single-GPU grouped computation
    assert par["bit_identical"] and par["max_abs_diff"] == 0.0 and par["ranks"] == 2 and par["rows_checked_per_rank"] > 0
