#!/usr/bin/env python3
"""Time the grouped MoE call (8 experts 4096->11008 by default) over a sweep of routed rows per expert
(product call fql_moe_fwd_f32 = pre-pass + grouped GEMM, hipGraph of 16 launches over rotating weight sets)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=4096); ap.add_argument("--n", type=int, default=11008)
ap.add_argument("--experts", type=int, default=8)
ap.add_argument("--rows", default="1,2,4,8,16,32,64,128,256")
ap.add_argument("--precision", default="exact")
ap.add_argument("--sets", type=int, default=4)
a = ap.parse_args()
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(0)
E = a.experts
sets = []
for _ in range(a.sets):
    w = torch.randn(E, a.n, a.k, device=dev, generator=g) * 0.02
    sets.append(fq.quantize_weights_moe(w))
    del w
wbytes = E * a.n * a.k // 2
for m in [int(b) for b in a.rows.split(",")]:
    T = m * E
    x = torch.randn(T, a.k, device=dev, generator=g)
    tpe = torch.full((E,), m, dtype=torch.int32, device=dev)
    offs = (torch.arange(E, device=dev, dtype=torch.int32) * m).contiguous()
    eid = torch.arange(E, device=dev, dtype=torch.int32).repeat_interleave(m)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in sets[:2]:
            ops.moe_forward(*s, x, eid, tpe, offs, precision=a.precision)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(16):
                ops.moe_forward(*sets[i % len(sets)], x, eid, tpe, offs, precision=a.precision)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 80 * 1e3
    print(f"rows/expert={m:4d} (T={T:5d}): {us:8.1f} us/call   {wbytes/us/1e6:7.2f} TB/s packed-weight   "
          f"{2.0*T*a.k*a.n/us/1e6:8.1f} TFLOP/s", flush=True)
