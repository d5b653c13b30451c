#!/usr/bin/env python3
"""Per-group scales along K and the bias epilogue: product calls timed in a
hipGraph over cold weights, next to the per-row call of the same shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops

dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    from fused_int4_amd import _native
    print("group i8 min rows", _native.lib().fql_tune_set_group_i8_min_rows(int(sys.argv[1])), "->", sys.argv[1])
K, N, G = 4096, 11008, 128
g = torch.Generator(device=dev).manual_seed(0)
NSETS = 36


def timed(fn_of_set, sets):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn_of_set(sets[0]); fn_of_set(sets[1])
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=st):
            for s in sets:
                fn_of_set(s)
        gr.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(3):
            gr.replay()
        e1.record(st)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (3 * len(sets)) * 1e3


row_sets, grp_sets = [], []
for i in range(NSETS):
    w = torch.randn(N, K, device=dev, generator=g) * 0.02
    row_sets.append(fq.quantize_weights(w))
    grp_sets.append(fq.quantize_weights(w, group_size=G))
bias = torch.randn(N, device=dev, generator=g)
for B in (8, 16, 32, 48, 64, 512):
    x = torch.randn(B, K, device=dev, generator=g)
    t_row = timed(lambda s: ops.linear_forward(x, *s), row_sets)
    t_bias = timed(lambda s: ops.linear_forward(x, *s, bias=bias), row_sets)
    t_grp = timed(lambda s: ops.linear_forward(x, *s), grp_sets)
    print(f"linear {K}->{N} B={B:4d}: per-row {t_row:8.1f} us | per-row + bias {t_bias:8.1f} us | per-group (g={G}) {t_grp:8.1f} us")

# grouped MoE, 8 experts (cold: 4 weight sets of 180 MB)
E = 8
moe_row, moe_grp = [], []
for i in range(4):
    P, S, Z, Pg, Sg, Zg = [], [], [], [], [], []
    for e in range(E):
        w = torch.randn(N, K, device=dev, generator=g) * 0.02
        p, s, z = fq.quantize_weights(w); P.append(p); S.append(s); Z.append(z)
        p, s, z = fq.quantize_weights(w, group_size=G); Pg.append(p); Sg.append(s); Zg.append(z)
    moe_row.append((torch.stack(P), torch.stack(S), torch.stack(Z)))
    moe_grp.append((torch.stack(Pg), torch.stack(Sg), torch.stack(Zg)))
for m in (8, 16, 32, 128):
    T = E * m
    x = torch.randn(T, K, device=dev, generator=g)
    tpe = torch.full((E,), m, dtype=torch.int32, device=dev)
    offs = torch.arange(E, dtype=torch.int32, device=dev) * m
    t_row = timed(lambda s: ops.moe_forward(s[0], s[1], s[2], x, None, tpe, offs), moe_row)
    t_grp = timed(lambda s: ops.moe_group_forward(s[0], s[1], s[2], x, tpe, offs), moe_grp)
    print(f"moe 8 x {K}->{N}, {m} rows per expert: per-row {t_row:8.1f} us | per-group (g={G}) {t_grp:8.1f} us")
