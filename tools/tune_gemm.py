#!/usr/bin/env python3
"""A/B the MFMA GEMM tile configurations on the GPU in ONE process (interleaved rounds), checking that
every configuration returns bit-identical outputs (the integer dot products are exact).

    python tools/tune_gemm.py [--workload moe|linear512|linear (--tokens rows)] [--precision exact|fast] [--cfgs 0,1,2] [--rounds 5]
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import fused_int4_amd as fq  # noqa: E402
from fused_int4_amd import ops, _native, routing as R  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="moe")
    ap.add_argument("--precision", default="exact")
    ap.add_argument("--cfgs", default="")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--sets", type=int, default=4)
    ap.add_argument("--tokens", type=int, default=512)
    ap.add_argument("--experts", type=int, default=8)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=11008)
    ap.add_argument("--routing", default="balanced")
    ap.add_argument("--top-k", type=int, default=2)
    ap.add_argument("--static-too", action="store_true", help="also time every configuration without scratch: static tile order, no residual pass (ids 2000 + cfg)")
    ap.add_argument("--alt-lib", default="", help="a second build of libfql_int4.so: every configuration is also timed through it (ids 1000 + cfg)")
    ap.add_argument("--plain-too", action="store_true", help="also time every configuration with the plain BN-wide column tiling instead of the balanced tile widths (ids 3000 + cfg)")
    ap.add_argument("--time-mismatched", action="store_true", help="also time configurations whose outputs differ (ablation builds)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _native.lib()
    tune = lib.fql_tune_gemm_i8_f32
    tune.restype = ctypes.c_int
    tune.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    alt = None
    if a.alt_lib:
        alt = ctypes.CDLL(os.path.abspath(a.alt_lib)).fql_tune_gemm_i8_f32
        alt.restype = ctypes.c_int
        alt.argtypes = tune.argtypes
    ncfg = lib.fql_tune_num_configs()
    prec_id = {"exact": 3, "fast": 2, "int8": 1, "fp8": 8}[a.precision]
    cfgs = [int(c) for c in a.cfgs.split(",")] if a.cfgs else [c for c in range(ncfg) if lib.fql_tune_is_config(c, prec_id)]
    base_cfgs = list(cfgs)
    if alt is not None:
        cfgs = cfgs + [1000 + c for c in base_cfgs]
    if a.static_too:
        cfgs = cfgs + [2000 + c for c in base_cfgs]
    if a.plain_too:
        cfgs = cfgs + [3000 + c for c in base_cfgs]
    prec = {"exact": 3, "fast": 2, "int8": 1, "fp8": 8}[a.precision]
    E, K, N = a.experts, a.hidden, a.ffn

    g = torch.Generator(device=dev).manual_seed(1)
    if a.workload == "moe":
        sets = []
        for i in range(a.sets):
            P, S, Z = [], [], []
            for e in range(E):
                w = torch.randn(N, K, device=dev, generator=g) * 0.02
                p, s, z = fq.quantize_weights(w)
                P.append(p); S.append(s); Z.append(z)
            sets.append((torch.stack(P), torch.stack(S), torch.stack(Z)))
        if a.routing == "balanced":
            route = R.balanced_routing(a.tokens, E, a.top_k, device=dev, seed=42)
        else:
            route = R.simulate_routing(a.tokens, E, a.top_k, "skewed", dev, 42)
        x_tok = torch.randn(a.tokens, K, device=dev, generator=g)
        x, tpe, offs, _ = R.dispatch_grouped(x_tok, route.expert_indices, E)
        x = x.contiguous()
        print("tokens_per_expert", tpe.tolist())
        tp, of, En = tpe.data_ptr(), offs.data_ptr(), E
    else:
        sets = []
        for i in range(a.sets):
            w = torch.randn(N, K, device=dev, generator=g) * 0.02
            sets.append(fq.quantize_weights(w))
        x = torch.randn(a.tokens if a.workload == "linear" else 512, K, device=dev, generator=g)
        tp, of, En = None, None, 1
    T = x.shape[0]
    limbs, delta, rowsum = ops.act_quant(x, precision=prec, tokens_per_expert=tpe if a.workload == "moe" else None,
                                          input_offsets=offs if a.workload == "moe" else None)
    stream = torch.cuda.current_stream().cuda_stream
    outs = {}
    flops = 2.0 * T * K * N
    wbytes = En * N * (K // 2)

    scratch = ops.gemm_scratch(prec, dev)

    def run(cfg, si, out):
        P, S, Z = sets[si % len(sets)]
        fn = alt if 1000 <= cfg < 2000 else tune
        sc = None if 2000 <= cfg < 3000 else scratch
        lib.fql_tune_set_balance_tiles(0 if cfg >= 3000 else 1)
        cfg = cfg % 1000
        rc = fn(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), P.data_ptr(), S.data_ptr(), Z.data_ptr(),
                  tp, of, out.data_ptr(), En, T, K, N, prec, stream,
                  None if sc is None else sc.data_ptr(), 0 if sc is None else sc.numel())
        assert rc == 0, (cfg, rc)

    ref = None
    ok = {}
    for c in cfgs:
        out = torch.zeros((T, N), dtype=torch.float32, device=dev)
        try:
            run(c, 0, out)
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print(f"cfg {c}: FAILED {exc}")
            ok[c] = False
            continue
        if ref is None:
            ref = out.clone()
            ok[c] = True
        else:
            same = torch.equal(out, ref)
            ok[c] = same
            if not same:
                print(f"cfg {c}: MISMATCH vs cfg {cfgs[0]} max|d|={(out - ref).abs().max().item():.3e}")
    times = {c: [] for c in cfgs if ok.get(c) or (a.time_mismatched and c in ok)}
    out = torch.empty((T, N), dtype=torch.float32, device=dev)
    for r in range(a.rounds):
        for c in times:
            run(c, 0, out); run(c, 1, out)                    # warm
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.iters)]
            for i, (e0, e1) in enumerate(evs):
                e0.record(); run(c, i + 2, out); e1.record()
            torch.cuda.synchronize()
            times[c].append(sorted(e0.elapsed_time(e1) for e0, e1 in evs)[len(evs) // 2])
    print(f"workload={a.workload} precision={a.precision} T={T} K={K} N={N}")
    for c, ts in times.items():
        ts = sorted(ts)
        med = ts[len(ts) // 2]
        print(f"cfg {c:4d}: median {med*1e3:8.1f} us  min {ts[0]*1e3:8.1f} us   {flops/med/1e9:8.1f} TFLOP/s alg   "
              f"{wbytes/med/1e6:7.1f} GB/s packed   bitexact={ok[c]}")


if __name__ == "__main__":
    main()
