#!/usr/bin/env python3
"""Benchmark of the fused INT4 hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py [--gpus N --steps K --warmup W] [--workload moe|linear512|linear1]

A *step* is one pass of the hot path over one batch of synthetic input, inputs resident in HBM:
  moe       (default, BASELINE.json configs[2]) QuantizedMoE 8 experts 4096->11008, 512 tokens top-2
            = 1024 routed rows pre-grouped by expert: activation pre-pass + grouped INT4 GEMM.
  linear512 (configs[1]) QuantizedLinear 4096->11008, batch 512.
  linear1   (configs[0] shape on the GPU) QuantizedLinear 4096->11008, batch 1 (GEMV kernel).
With --gpus N > 1 (launched under torch.distributed.run, one rank per GPU) the MoE workload is
expert-sharded: E/N experts and 512/N tokens per rank, RCCL all-to-all dispatch and combine
(strong scaling: the total work is the 1-GPU workload).

Rank 0 prints ONE JSON line.  `value` = algorithmic TFLOP/s of the whole job, 2*rows*K*N flops per
step (each multiply-add counted once, however many INT8 limbs the kernel spends on it).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_I8_PEAK_TOPS = 5000.0      # dense INT8 MFMA = 2x the ~2.5 PF bf16 dense rate (microarch guide, Matrix cores)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="moe", choices=["moe", "linear512", "linear1"])
    ap.add_argument("--precision", default="default", choices=["default", "exact", "fast", "int8", "fp8"])
    ap.add_argument("--routing", default="balanced", choices=["balanced", "skewed"])
    ap.add_argument("--weight-sets", type=int, default=4,
                    help="distinct weight copies rotated through so that consecutive steps cannot be served "
                         "from the 256 MB Infinity Cache (4 x 180 MB); 1 = warm-cache numbers")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--one-launch", type=int, default=-1, choices=[-1, 0, 1],
                    help="A/B switch (tuning hook fql_tune_set_fused): 1 = pre-pass as the GEMM kernel's first phase, 0 = two launches, "
                         "-1 = the library's default")
    ap.add_argument("--no-side-modes", action="store_true",
                    help="skip the opt-in precision / float16 side measurements (profiling runs: one kernel variant only)")
    ap.add_argument("--experts", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=11008)
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--top-k", type=int, default=2)
    return ap.parse_args()


def quantize_on_device(w):
    import fused_int4_amd as fq
    return fq.quantize_weights(w)


def make_weights(E, N, K, dev, seed):
    """randn(N,K)*0.02 per expert (reference: benchmark/moe_grouped_gemm/moe_int4_module.py:151-154),
    quantised per row with the package's quantize_weights."""
    g = torch.Generator(device=dev).manual_seed(seed)
    P, S, Z = [], [], []
    for _ in range(E):
        w = torch.randn(N, K, device=dev, generator=g) * 0.02
        p, s, z = quantize_on_device(w)
        P.append(p); S.append(s); Z.append(z)
        del w
    return torch.stack(P), torch.stack(S), torch.stack(Z)


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _timed_passes(fn, threads, budget_s, warm=3, reps=10):
    """SURVEY 8(d) protocol: `warm` untimed + up to `reps` timed passes (time.perf_counter), bounded by `budget_s`
    seconds of wall clock; returns (median, min, passes timed)."""
    torch.set_num_threads(threads)
    t_start = time.perf_counter()
    for _ in range(warm):
        fn()
        if time.perf_counter() - t_start > budget_s / 3:
            break
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_start > budget_s:
            break
    ts.sort()
    return ts[len(ts) // 2], ts[0], len(ts)


def _host_threads():
    """Threads for the all-core CPU figure: the cores this process may run on, capped at the pool's CPU share per GPU
    (16): a 256-thread torch pool on a 16-core share ran the same pass 5x SLOWER than one thread."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, int(os.environ.get("FQL_BENCH_CPU_THREADS", "16"))))


def cpu_baseline_moe(P, S, Z, x, tpe, offs, K, N, budget_s=24.0):
    """The oracle (CPU port of the reference's dequantize-then-matmul, python/quantize.py:176-202, applied per expert as
    benchmark/moe_grouped_gemm/moe_int4_module.py:123-125 does) timed on the host cores on a BOUNDED sample of the
    same workload: the first experts of the pass, as many as fit the time budget; all cores and one thread."""
    from oracle import oracle_torch as OT       # the oracle in the reference's own tensor ops (multi-threaded elementwise + F.linear)
    Pn, Sn, Zn, xn = P.cpu(), S.cpu(), Z.cpu(), x.cpu()
    tp, of = tpe.cpu().numpy(), offs.cpu().numpy()
    E = Pn.shape[0]
    cores = _host_threads()

    def sample(n_exp):
        def run():
            for e in range(n_exp):
                c, o = int(tp[e]), int(of[e])
                if c > 0:
                    OT.reference_quantized_linear(xn[o:o + c], Pn[e], Sn[e], Zn[e])
        return run
    rows = lambda n_exp: int(sum(int(tp[e]) for e in range(n_exp)))
    torch.set_num_threads(cores)
    t0 = time.perf_counter(); sample(1)(); t1 = time.perf_counter() - t0          # one expert, cold: sizes the sample
    n_all = max(1, min(E, int((budget_s * 0.6) / (13 * max(t1, 1e-3)))))
    med_a, min_a, n_a = _timed_passes(sample(n_all), cores, budget_s * 0.6)
    n_one = 1
    med_1, min_1, n_1 = _timed_passes(sample(n_one), 1, budget_s * 0.4, warm=1, reps=3)
    torch.set_num_threads(cores)
    val_a = 2.0 * rows(n_all) * K * N / med_a / 1e12
    val_1 = 2.0 * rows(n_one) * K * N / med_1 / 1e12
    return {"value": val_a, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"first {n_all} of {E} experts ({rows(n_all)} routed rows) per pass, dequantize_weights + F.linear per "
                      f"expert; {n_a} timed passes after warm-up, median {med_a*1e3:.1f} ms (min {min_a*1e3:.1f} ms)",
            "one_thread": {"value": val_1, "unit": "TFLOP/s", "cores": 1,
                           "sample": f"first {n_one} expert(s) ({rows(n_one)} rows), {n_1} timed passes, median {med_1*1e3:.1f} ms"},
            "cpu_model": _cpu_model(), "torch": torch.__version__, "os_cpu_count": os.cpu_count()}


def cpu_baseline_linear(p, s, z, x, K, N, budget_s=24.0):
    from oracle import oracle_torch as OT
    pn, sn, zn, xn = p.cpu(), s.cpu(), z.cpu(), x.cpu()
    if xn.dim() == 1:
        xn = xn[None]
    cores = _host_threads()
    run = lambda: OT.reference_quantized_linear(xn, pn, sn, zn)
    med_a, min_a, n_a = _timed_passes(run, cores, budget_s * 0.6)
    med_1, min_1, n_1 = _timed_passes(run, 1, budget_s * 0.4, warm=1, reps=5)
    torch.set_num_threads(cores)
    fl = 2.0 * xn.shape[0] * K * N
    return {"value": fl / med_a / 1e12, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"full passes of dequantize_weights + F.linear, batch {xn.shape[0]}: {n_a} timed after warm-up, "
                      f"median {med_a*1e3:.1f} ms (min {min_a*1e3:.1f} ms)",
            "one_thread": {"value": fl / med_1 / 1e12, "unit": "TFLOP/s", "cores": 1,
                           "sample": f"{n_1} timed passes, median {med_1*1e3:.1f} ms"},
            "cpu_model": _cpu_model(), "torch": torch.__version__, "os_cpu_count": os.cpu_count()}


def check_outputs(out, x, P, S, Z, tpe, offs, prec, rows_per_group=3):
    """A12 / VERDICT: compare the output of the call that gets timed with the oracle (float64 accumulation on the host,
    oracle/int4_oracle.c) on sampled rows of every group -- first, middle and last row -- BEFORE the timed loop.
    Returns the largest relative (Frobenius, per row) error; raises when it is outside the stated tolerance."""
    import numpy as np
    from oracle import c_oracle as C
    from oracle import oracle as O
    tol = {"default": 2e-6, "exact": 2e-6, "fast": 2e-4, "int8": 2.5e-2, "fp8": 1e-4}[prec]     # tests/helpers.py
    grouped = P.dim() == 3
    E = P.shape[0] if grouped else 1
    tp = tpe.cpu().numpy() if grouped else [x.shape[0]]
    of = offs.cpu().numpy() if grouped else [0]
    worst, checked = 0.0, 0
    picks = list(range(E)) if E <= 8 else sorted({0, 1, E // 2, E - 2, E - 1})
    for e in picks:
        c, o = int(tp[e]), int(of[e])
        if c <= 0:
            continue
        pe, se, ze = (t[e] if grouped else t for t in (P, S, Z))
        pe, se, ze = pe.cpu().numpy(), se.cpu().numpy(), ze.cpu().numpy()
        for r in sorted({o, o + c // 2, o + c - 1})[:rows_per_group]:
            xr = x[r].float().cpu().numpy()
            if prec == "fp8":                                  # kernel error on the same e4m3 activations
                xq, xs = O.quantize_activations_fp8(xr[None])
                ref = C.linear_f64acc(O.e4m3_decode(xq[0]), pe, se, ze) * np.float64(xs[0])
            else:
                ref = C.linear_f64acc(xr, pe, se, ze)
            got = out[r].float().cpu().numpy().astype(np.float64)
            err = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
            worst = max(worst, err)
            checked += 1
    if not (worst < tol):
        raise SystemExit(f"bench.py: output check FAILED: max relative error {worst:.3e} over {checked} sampled rows "
                         f"(tolerance {tol:g} for precision {prec!r})")
    return {"max_rel_err": worst, "rows_checked": checked, "tolerance": tol,
            "checker": "oracle/int4_oracle.c (float64 accumulation) on first / middle / last row of the sampled groups"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if a.one_launch >= 0:
        from fused_int4_amd import _native
        _native.lib().fql_tune_set_fused(a.one_launch)
    # Rehearsal mode (FQL_BENCH_BACKEND=gloo): several ranks share the visible card(s) and the collectives go
    # through host memory over gloo -- exercises this file's multi-rank path on a 1-GPU box; never a measurement.
    rehearsal = os.environ.get("FQL_BENCH_BACKEND", "nccl") == "gloo"
    dev_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
            import fused_int4_amd.ep as _ep
            _real_a2a = dist.all_to_all_single

            def _a2a_through_host(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
                o = torch.empty(output.shape, dtype=output.dtype)
                _real_a2a(o, input.cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
                output.copy_(o)
            _ep.dist.all_to_all_single = _a2a_through_host
        else:
            dist.init_process_group("nccl", device_id=dev)

    def all_gather_cat(t):      # [n, ...] of every rank -> [world * n, ...] (rank-major)
        src = t.contiguous().cpu() if rehearsal else t.contiguous()
        parts = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(parts, src)
        return torch.cat(parts).to(t.device)

    def all_reduce_sum(t):
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)

    def all_reduce_max(t):
        if rehearsal:
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX)
            t.copy_(h)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)

    import fused_int4_amd as fq
    from fused_int4_amd import ops, routing as R
    from fused_int4_amd.ep import ExpertParallelMoE

    E, K, N = a.experts, a.hidden, a.ffn
    prec = a.precision
    limbs = {"fast": 2, "int8": 1, "fp8": 1}.get(prec, 3)
    extra = {}

    def barrier():
        if world > 1:
            dist.barrier()

    # ------------------------------------------------------------------ build the workload
    if a.workload == "moe" and world == 1:
        T = a.tokens * a.top_k
        sets = [make_weights(E, N, K, dev, 42 + i) for i in range(max(1, a.weight_sets))]
        route = (R.balanced_routing if a.routing == "balanced" else
                 lambda t, e, k, device, seed: R.simulate_routing(t, e, k, "skewed", device, seed))(
            a.tokens, E, a.top_k, device=dev, seed=42)
        torch.manual_seed(42)
        x_tok = torch.randn(a.tokens, K, device=dev)
        x, tpe, offs, _ = R.dispatch_grouped(x_tok, route.expert_indices, E)
        x = x.contiguous()
        rows = T
        step_i = [0]

        def step():
            P, S, Z = sets[step_i[0] % len(sets)]
            step_i[0] += 1
            return ops.moe_forward(P, S, Z, x, None, tpe, offs, precision=prec)

        def phases(i):   # the same two kernels through the two-phase C entry points, for per-kernel timing
            P, S, Z = sets[i % len(sets)]
            return P, S, Z, tpe, offs
        flops = 2.0 * rows * K * N
        weight_bytes = E * N * (K // 2)
        all_bytes = weight_bytes + 2 * E * N * 4 + rows * K * 4 + rows * N * 4
        workload = f"QuantizedMoE {E} experts {K}->{N} top-{a.top_k}, batch {a.tokens} ({rows} routed rows, {a.routing})"
        parallelism = "1 GPU"
    elif a.workload == "moe":
        assert E % world == 0 and a.tokens % world == 0
        Pfull = None
        # every rank builds only its own experts' weights (same seeds as the 1-GPU run would use per expert block)
        EL = E // world
        g = torch.Generator(device=dev).manual_seed(42)
        sets = [make_weights(EL, N, K, dev, 42 + 1000 * rank + i) for i in range(max(1, a.weight_sets))]
        t_local = a.tokens // world
        route = R.balanced_routing(a.tokens, E, a.top_k, device=dev, seed=42)
        idx = route.expert_indices[rank * t_local:(rank + 1) * t_local].contiguous()
        wts = route.expert_weights[rank * t_local:(rank + 1) * t_local].contiguous()
        torch.manual_seed(42 + rank)
        x = torch.randn(t_local, K, device=dev)
        eps = [ExpertParallelMoE(E, P, S, Z, precision=prec) for (P, S, Z) in sets]
        step_i = [0]

        def step():
            ep = eps[step_i[0] % len(eps)]
            step_i[0] += 1
            return ep(x, idx, wts)
        phases = None
        ep_objects = eps
        rows = a.tokens * a.top_k
        flops = 2.0 * rows * K * N
        weight_bytes = E * N * (K // 2)
        all_bytes = weight_bytes + 2 * E * N * 4 + rows * K * 4 + rows * N * 4
        workload = (f"QuantizedMoE {E} experts sharded {EL}/GPU over {world} GPUs, {K}->{N} top-{a.top_k}, "
                    f"batch {a.tokens}, RCCL all-to-all dispatch/combine")
        parallelism = f"ep{world}"
    else:
        B = 512 if a.workload == "linear512" else 1
        # one weight set is 22.5 MB: rotate through enough of them (3 x the 256 MB Infinity Cache) that a step can
        # find none of its weights in a cache, unless warm numbers were asked for (--weight-sets 1)
        n_sets = 1 if a.weight_sets == 1 else max(a.weight_sets, -(-3 * 256 * 2**20 // (N * (K // 2))))
        sets = [tuple(t[0] for t in make_weights(1, N, K, dev, 42 + i)) for i in range(n_sets)]
        torch.manual_seed(42)
        x = torch.randn(B, K, device=dev)
        rows = B
        step_i = [0]

        def step():
            p, s, z = sets[step_i[0] % len(sets)]
            step_i[0] += 1
            return ops.linear_forward(x, p, s, z, precision=prec)

        def phases(i):
            p, s, z = sets[i % len(sets)]
            return p, s, z, None, None
        if B <= 4:
            phases = None
        flops = 2.0 * B * K * N
        weight_bytes = N * (K // 2)
        all_bytes = weight_bytes + 2 * N * 4 + B * K * 4 + B * N * 4
        workload = f"QuantizedLinear {K}->{N}, batch {B}"
        parallelism = "replicas only" if world > 1 else "1 GPU"

    # ------------------------------------------------------------------ timed region (the contract)
    # Before the W warm-up steps: keep the GPU busy with the same call for ~0.25 s.  The workload construction above ends
    # with host-side work, the device drops into a low-power state meanwhile, and climbing out of it took 30-70 ms on the
    # MI355X boxes of this pool -- longer than W short steps, so it used to land inside the timed region.
    if world == 1:
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < 0.25:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
    else:
        # every rank must make the SAME number of (collective) steps: a fixed count, never a wall-clock loop
        for _ in range(100):
            step()
        torch.cuda.synchronize()
    for _ in range(a.warmup):
        step()
    barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize(); barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        all_reduce_max(tmax)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / a.steps * 1e3
    total_flops = flops * (world if (a.workload != "moe" and world > 1) else 1)
    value = total_flops / (ms_per_step * 1e-3) / 1e12

    # ------------------------------------------------------------------ expert-parallel parity (outside the timed region)
    # The first rows of every rank's output must be, bit for bit (top-k <= 2: one addition per output), what ONE GPU's
    # grouped computation gives for the same tokens: every rank runs the single-GPU grouped call on the sampled rows
    # that route to ITS experts, the un-weighted rows are summed over the ranks (one contributor per row: exact) and
    # combined by the same kernel the expert-parallel step uses.  A mismatch aborts the run without a JSON line.
    if world > 1 and a.workload == "moe":
        ns = min(8, t_local)
        EL = E // world

        all_x, all_idx, all_w = all_gather_cat(x[:ns]), all_gather_cat(idx[:ns]), all_gather_cat(wts[:ns])
        y_ep = eps[0](x, idx, wts)[:ns].contiguous()                       # one more step on weight set 0 (a collective)
        flat_e = all_idx.reshape(-1).long()
        mine = torch.nonzero(torch.div(flat_e, EL, rounding_mode="floor") == rank).reshape(-1)
        rows_unw = torch.zeros((flat_e.numel(), N), dtype=torch.float32, device=dev)
        if mine.numel():
            le = (flat_e[mine] - rank * EL)
            order = torch.argsort(le, stable=True)
            grouped = all_x.repeat_interleave(a.top_k, 0).index_select(0, mine.index_select(0, order)).contiguous()
            tpe_l = torch.bincount(le, minlength=EL).to(torch.int32)
            offs_l = (torch.cumsum(tpe_l, 0) - tpe_l).to(torch.int32)
            Pl, Sl, Zl = sets[0]
            y_l = ops.moe_forward(Pl, Sl, Zl, grouped, None, tpe_l, offs_l, precision=prec)
            rows_unw[mine.index_select(0, order)] = y_l
        all_reduce_sum(rows_unw)
        pos = torch.arange(flat_e.numel(), dtype=torch.int32, device=dev)
        ref = ops.combine(rows_unw, pos, all_w.float())[rank * ns:(rank + 1) * ns]
        same = bool(torch.equal(ref, y_ep))
        diff = float((ref - y_ep).abs().max())
        flags = torch.tensor([0 if same else 1, 0], device=dev, dtype=torch.float64)
        flags[1] = diff
        all_reduce_max(flags)
        extra["parity_vs_single_gpu"] = {"rows_checked_per_rank": ns, "ranks": world, "bit_identical": float(flags[0]) == 0.0,
                                         "max_abs_diff": float(flags[1]),
                                         "how": "sampled rows of every rank's expert-parallel output vs the single-GPU grouped call on "
                                                "the owning rank + the same combine kernel"}
        if a.top_k <= 2 and not extra["parity_vs_single_gpu"]["bit_identical"]:
            raise SystemExit(f"expert-parallel output differs from the single-GPU grouped computation: max |d| = {float(flags[1]):.3e}")

    # ------------------------------------------------------------------ expert-parallel phases (outside the timed region)
    ep_roofline = None
    if world > 1 and a.workload == "moe":
        for ep in ep_objects:
            ep.record_phases = True
        for _ in range(max(4, min(20, a.steps))):
            step()
        torch.cuda.synchronize()
        acc = {}
        for ep in ep_objects:
            for k, v in ep.phase_times_ms().items():
                acc.setdefault(k, []).append(v)
            ep.record_phases = False
        mine = {k: sum(v) / len(v) for k, v in acc.items()}
        names = sorted(mine)
        t = torch.tensor([mine[k] for k in names], dtype=torch.float64, device=dev)
        all_reduce_max(t)
        extra["ep_phases_ms_max_over_ranks"] = {k: float(v) for k, v in zip(names, t.tolist())}
        t_g = extra["ep_phases_ms_max_over_ranks"].get("regroup_and_grouped_gemm")
        if t_g:
            ach = flops / world / (t_g * 1e-3) / 1e12
            ep_roofline = {"bound": "mfma", "kernel": "grouped INT4 GEMM of one rank's local experts (gemm_w4_kernel / gemm_i8_* by rows per expert)", "achieved": ach,
                           "peak": MFMA_I8_PEAK_TOPS, "unit": "TFLOP/s", "frac": ach / MFMA_I8_PEAK_TOPS, "traffic": None,
                           "note": "per GPU: this rank's share of the algorithmic flops over its regroup + pre-pass + grouped "
                                   "GEMM phase (GPU events, max over ranks); the step itself is all-to-all latency bound"}
        extra["ep_phases_note"] = ("GPU-event time between phase boundaries of one expert-parallel step (dispatch / grouped GEMM / "
                                   "combine reported separately, SURVEY 8d config 4); includes host gaps inside a phase")

    # ------------------------------------------------------------------ per-kernel durations (HIP events on the launch stream)
    roofline = ep_roofline
    if phases is not None and world == 1:
        # How the GEMM kernel's duration is taken (HIP events on the launch stream, right after the timed region):
        #   * an event pair around ONE launch also times the launch path (marker packets either side of the dispatch:
        #     ~15 us on this stack -- the "kernel" would come out longer than the whole step), so a pair brackets a BATCH;
        #   * the batches alternate the two kernels exactly as a step does, and start right behind a wake-up loop: after
        #     ANY idle gap of a few ms (the allocations below are one) the next ~15 ms of launches run slow (134 -> 169 us
        #     per GEMM launch in the rocprofv3 trace; profiles/r02_duty_cycle_check.txt);
        #   * GEMM = (batch of [pre-pass, GEMM] pairs  -  batch of pre-passes) / launches.  Kernel-boundary gaps stay inside;
        #     the rocprofv3 average of the same kernel over the same command (profiles/) is the cross-check.
        BATCH = 8
        nb = max(4, (max(a.steps, 20) + BATCH - 1) // BATCH)
        n = nb * BATCH
        outs = torch.empty((rows, N), dtype=torch.float32, device=dev)
        Pw, Sw, Zw, tp, of = phases(0)
        bufs = ops.act_quant(x, precision=prec, tokens_per_expert=tp, input_offsets=of)       # allocate once
        ev_p = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(nb)]
        ev_s = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(nb)]
        # the allocations above idled the GPU for tens of milliseconds: after any idle gap of a few ms the first ~15 ms of
        # work run at a lower clock (150-170 us per GEMM launch instead of 130 in the rocprofv3 trace) -- same wake-up as
        # before the timed region
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < 0.06:
            for i in range(8):
                Pw, Sw, Zw, tp, of = phases(i)
                lm, dl, rs = ops.act_quant(x, precision=prec, tokens_per_expert=tp, input_offsets=of, out=bufs)
                ops.gemm_i8(lm, dl, rs, Pw, Sw, Zw, tp, of, precision=prec, out=outs)
            torch.cuda.synchronize()
        for b in range(nb):
            ev_s[b][0].record()
            for i in range(BATCH):
                Pw, Sw, Zw, tp, of = phases(b * BATCH + i)
                lm, dl, rs = ops.act_quant(x, precision=prec, tokens_per_expert=tp, input_offsets=of, out=bufs)
                ops.gemm_i8(lm, dl, rs, Pw, Sw, Zw, tp, of, precision=prec, out=outs)
            ev_s[b][1].record()
            ev_p[b][0].record()
            for i in range(BATCH):
                ops.act_quant(x, precision=prec, tokens_per_expert=tp, input_offsets=of, out=bufs)
            ev_p[b][1].record()
        torch.cuda.synchronize()
        pre = sorted(e[0].elapsed_time(e[1]) / BATCH for e in ev_p)
        both = sorted(e[0].elapsed_time(e[1]) / BATCH for e in ev_s)
        # a preempted host thread (shared box) shows up as GPU idle time inside a batch: batches above 1.5x the median are
        # host stalls, not kernel time -- they are left out of the averages, and counted.
        keep = [g for g in both if g <= 1.5 * both[nb // 2]]
        keep_p = [q for q in pre if q <= 1.5 * pre[nb // 2]]
        pre_ms = sum(keep_p) / len(keep_p)
        gemm_ms = sum(keep) / len(keep) - pre_ms
        extra.update({"gemm_kernel_ms_avg": gemm_ms, "gemm_kernel_ms_median": both[nb // 2] - pre[nb // 2],
                      "gemm_kernel_ms_min": both[0] - pre[0],
                      "gemm_kernel_samples_dropped_as_host_stalls": nb - len(keep),
                      "gemm_kernel_launches_timed": n, "gemm_kernel_launches_per_event_pair": BATCH,
                      "gemm_kernel_ms_how": "HIP events: (batch of [pre-pass, GEMM] launch pairs - batch of pre-pass launches) / launches",
                      "act_quant_ms_avg": pre_ms, "act_quant_ms_median": pre[nb // 2],
                      # the same two launches as a step, per pair, from the event-bracketed batches (median and min)
                      "step_ms_median_events": both[nb // 2], "step_ms_min_events": both[0]})
        mfma_achieved = flops / (gemm_ms * 1e-3) / 1e12
        hbm_achieved = weight_bytes / (gemm_ms * 1e-3) / 1e9
        mfma_floor_ms = flops * limbs / (MFMA_I8_PEAK_TOPS * 1e12) * 1e3
        hbm_floor_ms = weight_bytes / (HBM_PEAK_GBPS * 1e9) * 1e3
        traffic = None
        tfile = os.path.join(ROOT, "profiles", f"pmc_traffic_{a.workload}.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                shape = tj.get("shape", {})
                same = (shape.get("experts") == a.experts and shape.get("hidden") == K and shape.get("ffn") == N
                        and shape.get("rows") == rows and prec in ("default", "exact") and a.routing == "balanced")
                traffic = tj.get("hbm_bytes_per_launch") if same else None     # PMC figure of THIS shape only
            except Exception:
                traffic = None
        kname = "gemm_i8_kernel"
        try:    # the kernel family the library picks for this shape (tuning hook; ids as in fql_tune_gemm_i8_f32)
            from fused_int4_amd import _native
            cfg_id = _native.lib().fql_tune_chosen_cfg(ops._precision(prec), a.experts if a.workload == "moe" else 1, rows, K, N,
                                                       1 if a.workload == "moe" else 0)
            kname = ("gemm_w4_kernel" if cfg_id >= 300 else "gemm_i8_rows16_kernel" if cfg_id >= 200 else
                     "gemm_i8_rows32_kernel" if cfg_id >= 100 else "gemm_i8_kernel")
            extra["gemm_tile_configuration"] = cfg_id
        except Exception:
            pass
        if mfma_floor_ms >= hbm_floor_ms:
            roofline = {"bound": "mfma", "kernel": kname, "achieved": mfma_achieved, "peak": MFMA_I8_PEAK_TOPS,
                        "unit": "TFLOP/s", "frac": mfma_achieved / MFMA_I8_PEAK_TOPS, "traffic": traffic,
                        "note": f"algorithmic flops 2*rows*K*N counted once; the kernel issues {limbs} INT8 MFMA passes "
                                f"(one per activation limb), so frac <= 1/{limbs} by construction"}
        else:
            roofline = {"bound": "hbm", "kernel": kname, "achieved": hbm_achieved, "peak": HBM_PEAK_GBPS,
                        "unit": "GB/s", "frac": hbm_achieved / HBM_PEAK_GBPS, "traffic": traffic}
        extra["roofline_hbm_packed_weights"] = {"achieved": hbm_achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                                "frac": hbm_achieved / HBM_PEAK_GBPS, "bytes": weight_bytes}
        extra["roofline_mfma_i8"] = {"achieved": mfma_achieved, "peak": MFMA_I8_PEAK_TOPS, "unit": "TFLOP/s",
                                     "frac": mfma_achieved / MFMA_I8_PEAK_TOPS, "issued_frac": mfma_achieved * limbs / MFMA_I8_PEAK_TOPS}
    elif world == 1:   # GEMV: one ~10 us kernel per step.  Eager launches from Python are host-bound at that
        # size, so the kernel's average duration is taken from a hipGraph of `n` back-to-back launches
        # (each still pays the ~1.5 us dependent-launch boundary), timed with HIP events on the stream.
        n = max(20, min(a.steps, 200))
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            step()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                for _ in range(n):
                    step()
            graph.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 5
            e0.record(stream)
            for _ in range(reps):
                graph.replay()
            e1.record(stream)
        torch.cuda.synchronize()
        k_ms = e0.elapsed_time(e1) / (reps * n)
        ach = (weight_bytes + 2 * N * 4 + rows * K * 4) / (k_ms * 1e-3) / 1e9
        roofline = {"bound": "hbm", "kernel": "gemv_kernel", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                    "note": "bytes = the reference's own model (benchmark/run_benchmark.py:222): N*K/2 + 8N + 4K"}
        extra.update({"gemv_kernel_ms_avg_graph": k_ms, "graph_launches": n})

    # ------------------------------------------------------------------ the opt-in reduced-limb modes, for the record
    if world == 1 and a.workload == "moe" and prec in ("default", "exact") and not a.no_side_modes:
        def time_mode(mode):
            def step_mode():
                P, S, Z = sets[step_i[0] % len(sets)]
                step_i[0] += 1
                return ops.moe_forward(P, S, Z, x, None, tpe, offs, precision=mode)
            for _ in range(5):
                step_mode()
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            nf = max(10, a.steps // 4)
            for _ in range(nf):
                step_mode()
            torch.cuda.synchronize()
            return (time.perf_counter() - tf0) / nf * 1e3
        ms_fast = time_mode("fast")
        extra["fast_mode_2_limbs"] = {"ms_per_step": ms_fast, "value": flops / (ms_fast * 1e-3) / 1e12, "unit": "TFLOP/s",
                                      "note": "FQL_PRECISION_FAST: ~3e-5 relative error (north-star bound 1e-3); not the headline"}
        x16 = x.half()

        def time_f16():
            def step16():
                P, S, Z = sets[step_i[0] % len(sets)]
                step_i[0] += 1
                return ops.moe_forward_any(P, S, Z, x16, None, tpe, offs, precision=prec)
            for _ in range(5):
                step16()
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            nf = max(10, a.steps // 4)
            for _ in range(nf):
                step16()
            torch.cuda.synchronize()
            return (time.perf_counter() - tf0) / nf * 1e3
        ms_16 = time_f16()
        extra["float16_io"] = {"ms_per_step": ms_16, "value": flops / (ms_16 * 1e-3) / 1e12, "unit": "TFLOP/s",
                               "note": "float16 rows in, float16 out through fql_moe_fwd (widening / rounding fused into the "
                                       "kernels; same limbs and accumulation as the headline); not the headline"}
        ms_i8 = time_mode("int8")
        extra["int8_mode_1_limb"] = {"ms_per_step": ms_i8, "value": flops / (ms_i8 * 1e-3) / 1e12, "unit": "TFLOP/s",
                                     "hbm_GBps_packed_weights_e2e": weight_bytes / (ms_i8 * 1e-3) / 1e9,
                                     "note": "FQL_PRECISION_INT8: 8-bit activations per row, 1-2e-2 relative error (outside the 1e-3 claim); not the headline"}
        # the literal MoEINT4 API: ONE expert per row (T = tokens rows): half the rows, same weights
        if a.top_k == 2 and a.routing == "balanced" and a.tokens % E == 0:
            m1 = a.tokens // E
            tpe1 = torch.full((E,), m1, dtype=torch.int32, device=dev)
            offs1 = torch.arange(E, dtype=torch.int32, device=dev) * m1
            x1 = x[:a.tokens].contiguous()

            def step_top1():
                P, S, Z = sets[step_i[0] % len(sets)]
                step_i[0] += 1
                return ops.moe_forward(P, S, Z, x1, None, tpe1, offs1, precision=prec)
            for _ in range(5):
                step_top1()
            torch.cuda.synchronize()
            tf0 = time.perf_counter()
            nf = max(10, a.steps // 4)
            for _ in range(nf):
                step_top1()
            torch.cuda.synchronize()
            ms_1 = (time.perf_counter() - tf0) / nf * 1e3
            extra["single_assignment_top1"] = {"ms_per_step": ms_1, "rows": a.tokens, "value": flops / 2 / (ms_1 * 1e-3) / 1e12, "unit": "TFLOP/s",
                                               "hbm_GBps_packed_weights_e2e": weight_bytes / (ms_1 * 1e-3) / 1e9,
                                               "note": "one expert per row (the MoEINT4 call as the reference's harness makes it); not the headline"}

    # ------------------------------------------------------------------ side measurement: the reference's DEFAULT routing
    # distribution ("skewed", benchmark/run_moe_benchmark.py:106; routing.py:26-93) on the same weights and step
    if (a.workload == "moe" and world == 1 and a.routing == "balanced" and prec in ("default", "exact")
            and not a.no_side_modes):
        route_s = R.simulate_routing(a.tokens, E, a.top_k, "skewed", dev, 42)
        xs, tpe_s, offs_s, _ = R.dispatch_grouped(x_tok, route_s.expert_indices, E)
        xs = xs.contiguous()

        def step_skew():
            P, S, Z = sets[step_i[0] % len(sets)]
            step_i[0] += 1
            return ops.moe_forward(P, S, Z, xs, None, tpe_s, offs_s, precision=prec)
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < 0.1:
            for _ in range(8):
                step_skew()
            torch.cuda.synchronize()
        ns = max(20, a.steps)
        torch.cuda.synchronize()
        tf0 = time.perf_counter()
        for _ in range(ns):
            step_skew()
        torch.cuda.synchronize()
        ms_s = (time.perf_counter() - tf0) / ns * 1e3
        out_s = step_skew()
        torch.cuda.synchronize()
        Ps, Ss, Zs = sets[(step_i[0] - 1) % len(sets)]
        chk = check_outputs(out_s, xs, Ps, Ss, Zs, tpe_s, offs_s, prec)
        extra["skewed_routing"] = {"ms_per_step": ms_s, "tokens_per_expert": tpe_s.tolist(), "steps": ns,
                                   "value": flops / (ms_s * 1e-3) / 1e12, "unit": "TFLOP/s",
                                   "max_rel_err": chk["max_rel_err"],
                                   "note": "the reference's default 'skewed' routing on the same weights; ratio to the headline step in "
                                           "'vs_headline_step' is filled in below; not the headline"}
        del out_s

    # ------------------------------------------------------------------ side measurements: the weight-streaming shapes
    # (HBM-bound: the packed weights are read once per step whatever the row count).  decode32: 32 tokens, top-2 -> 8 rows
    # per expert on the same 8 x (4096 -> 11008) weights; linear1: QuantizedLinear 4096 -> 11008 at batch 1 (configs[0] on
    # the GPU), walking the experts' matrices of every weight set so that no call finds its weights in a cache.
    if (a.workload == "moe" and world == 1 and a.routing == "balanced" and prec in ("default", "exact")
            and not a.no_side_modes and a.tokens != 32):
        def time_side(fn, n):
            t_wake = time.perf_counter()
            while time.perf_counter() - t_wake < 0.05:
                for _ in range(8):
                    fn()
                torch.cuda.synchronize()
            tf0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - tf0) / n * 1e3
        rows_d = 32 * a.top_k
        md = max(1, rows_d // E)
        tpe_d = torch.full((E,), md, dtype=torch.int32, device=dev)
        offs_d = torch.arange(E, dtype=torch.int32, device=dev) * md
        xd = torch.randn(md * E, K, device=dev)

        def step_decode():
            P, S, Z = sets[step_i[0] % len(sets)]
            step_i[0] += 1
            return ops.moe_forward(P, S, Z, xd, None, tpe_d, offs_d, precision=prec)
        ms_d = time_side(step_decode, max(50, a.steps))
        out_d = step_decode()
        torch.cuda.synchronize()
        Pd, Sd, Zd = sets[(step_i[0] - 1) % len(sets)]
        chk_d = check_outputs(out_d, xd, Pd, Sd, Zd, tpe_d, offs_d, prec)
        extra["decode32"] = {"ms_per_step": ms_d, "rows_per_expert": md, "steps": max(50, a.steps),
                             "hbm_GBps_packed_weights_e2e": weight_bytes / (ms_d * 1e-3) / 1e9,
                             "hbm_frac_e2e": weight_bytes / (ms_d * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                             "max_rel_err": chk_d["max_rel_err"],
                             "note": "32 tokens, top-2: 8 rows per expert, pre-pass + GEMM; bytes = the packed weights only; not the headline"}
        del out_d
        x1l = torch.randn(1, K, device=dev)

        def step_linear1():
            P, S, Z = sets[(step_i[0] // E) % len(sets)]
            e = step_i[0] % E
            step_i[0] += 1
            return ops.linear_forward(x1l, P[e], S[e], Z[e], precision=prec)
        ms_l = time_side(step_linear1, max(100, a.steps))
        out_l = step_linear1()
        torch.cuda.synchronize()
        el = (step_i[0] - 1) % E
        Pl1, Sl1, Zl1 = sets[((step_i[0] - 1) // E) % len(sets)]
        chk_l = check_outputs(out_l, x1l, Pl1[el], Sl1[el], Zl1[el], None, None, prec)
        b1 = N * K // 2 + 8 * N + 4 * K
        extra["linear1"] = {"ms_per_step": ms_l, "steps": max(100, a.steps),
                            "hbm_GBps": b1 / (ms_l * 1e-3) / 1e9, "hbm_frac": b1 / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                            "max_rel_err": chk_l["max_rel_err"],
                            "note": "QuantizedLinear 4096 -> 11008 at batch 1 over the experts' matrices in turn (cold weights); "
                                    "bytes = the reference's model N*K/2 + 8N + 4K (benchmark/run_benchmark.py:222); not the headline"}
        del out_l

    # ------------------------------------------------------------------ output check of the timed call (A12 / VERDICT r1)
    # the same call once more, compared with the oracle on sampled rows of every group; a wrong result fails the run
    if world == 1 and os.environ.get("FQL_BENCH_SKIP_CHECK") != "1":
        out0 = step()
        torch.cuda.synchronize()
        if a.workload == "moe":
            P0, S0, Z0 = sets[(step_i[0] - 1) % len(sets)]
            extra["output_check"] = check_outputs(out0, x, P0, S0, Z0, tpe, offs, prec)
        else:
            p0, s0, z0 = sets[(step_i[0] - 1) % len(sets)]
            extra["output_check"] = check_outputs(out0 if out0.dim() == 2 else out0[None], x if x.dim() == 2 else x[None],
                                                  p0, s0, z0, None, None, prec)
        extra["max_rel_err"] = extra["output_check"]["max_rel_err"]
        del out0

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only)
    cpu = None
    if world == 1 and not a.no_cpu_baseline:
        if a.workload == "moe":
            P, S, Z = sets[0]
            cpu = cpu_baseline_moe(P, S, Z, x, tpe, offs, K, N)
        else:
            p, s, z = sets[0]
            cpu = cpu_baseline_linear(p, s, z, x, K, N)

    if "skewed_routing" in extra:
        extra["skewed_routing"]["vs_headline_step"] = extra["skewed_routing"]["ms_per_step"] / ms_per_step
    if rank == 0:
        line = {
            "metric": "fused INT4 GEMM effective TFLOP/s (ms, HBM GB/s and roofline fractions alongside)",
            "value": value, "unit": "TFLOP/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong" if a.workload == "moe" else "weak",
            "vs_baseline": None, "dtype": ("f32" if a.workload == "linear1" else "fp8" if prec == "fp8" else "i8"), "data": "synthetic",
            "config": {"workload": workload, "precision": prec, "activation_limbs": limbs,
                       "parallelism": parallelism,
                       "cache": (f"cold: {len(sets)} weight sets rotated" if len(sets) > 1 else "warm: one weight set")},
            "algorithmic_flops_per_step": flops, "packed_weight_bytes": weight_bytes, "all_bytes": all_bytes,
            "hbm_GBps_packed_weights_e2e": weight_bytes / (ms_per_step * 1e-3) / 1e9,
            "roofline": roofline, "cpu_baseline": cpu,
        }
        line.update(extra)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
