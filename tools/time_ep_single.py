#!/usr/bin/env python3
"""Single-GPU cost of the expert-parallel wrapper's bookkeeping (world size 1: sort by expert, gather, grouped
GEMM, un-sort, weighted sum) against the bare grouped call on pre-grouped rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops, routing as R
from fused_int4_amd.ep import ExpertParallelMoE
dev = torch.device("cuda:0"); E, K, N, T, topk = 8, 4096, 11008, 512, 2
g = torch.Generator(device=dev).manual_seed(0)
P, S, Z = [], [], []
for e in range(E):
    p, s, z = fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02)
    P.append(p); S.append(s); Z.append(z)
P, S, Z = torch.stack(P), torch.stack(S), torch.stack(Z)
route = R.balanced_routing(T, E, topk, device=dev, seed=42)
x = torch.randn(T, K, device=dev, generator=g)
ep = ExpertParallelMoE(E, P, S, Z)
xg, tpe, offs, _ = R.dispatch_grouped(x, route.expert_indices, E)
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print(f"bare grouped call on pre-grouped rows : {timeit(lambda: ops.moe_forward(P, S, Z, xg, None, tpe, offs)):8.1f} us")
print(f"expert-parallel wrapper, world size 1 : {timeit(lambda: ep(x, route.expert_indices, route.expert_weights)):8.1f} us")
ref = ep(x, route.expert_indices, route.expert_weights)
ep.fold_weights = True
got = ep(x, route.expert_indices, route.expert_weights)
print(f"... routing weights folded into the GEMM epilogue, combine = gather-add : "
      f"{timeit(lambda: ep(x, route.expert_indices, route.expert_weights)):8.1f} us   bit-identical={torch.equal(ref, got)}")
ep.fold_weights = False
for rep in range(2):
    print(f"(repeat) weighted combine {timeit(lambda: ep(x, route.expert_indices, route.expert_weights)):8.1f} us", end="   ")
    ep.fold_weights = True
    print(f"folded {timeit(lambda: ep(x, route.expert_indices, route.expert_weights)):8.1f} us")
    ep.fold_weights = False
