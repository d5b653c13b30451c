#!/usr/bin/env python3
"""Time QuantizedLinear 4096->11008 (or --k/--n) over a sweep of batch sizes (product call; a hipGraph of one launch per weight set, the sets 3 x the Infinity Cache in total: cold weights)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops
ap = argparse.ArgumentParser()
ap.add_argument("--k", type=int, default=4096); ap.add_argument("--n", type=int, default=11008)
ap.add_argument("--batches", default="1,2,4,5,8,16,32,64,128,256,512,1024")
ap.add_argument("--precision", default="exact")
ap.add_argument("--gemv-max", type=int, default=-1, help="tuning hook: batches up to this many rows take the GEMV kernel")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.gemv_max >= 0:
    from fused_int4_amd import _native
    print("gemv max rows", _native.lib().fql_tune_set_gemv_max_rows(a.gemv_max), "->", a.gemv_max)
g = torch.Generator(device=dev).manual_seed(0)
# enough weight sets that a launch finds none of its weights in the 256 MB Infinity Cache (3 x its size in rotation)
NSETS = max(8, -(-3 * 256 * 2**20 // (a.n * a.k // 2)))
sets = [fq.quantize_weights(torch.randn(a.n, a.k, device=dev, generator=g) * 0.02) for _ in range(NSETS)]
wbytes = a.n * a.k // 2
for B in [int(b) for b in a.batches.split(",")]:
    x = torch.randn(B, a.k, device=dev, generator=g)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for s in sets[:2]:
            ops.linear_forward(x, *s, precision=a.precision)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for i in range(NSETS):
                ops.linear_forward(x, *sets[i], precision=a.precision)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(5):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / (5 * NSETS) * 1e3
    print(f"B={B:5d}: {us:8.1f} us/call   {wbytes/us/1e6:7.2f} TB/s packed-weight   {2.0*B*a.k*a.n/us/1e6:8.1f} TFLOP/s")
