#!/usr/bin/env python3
"""Timeline of ONE hot-path step (pre-pass + GEMM) at BASELINE configs[2] under sustained load, from a -DFQL_TRACE build
(make -C .../csrc trace): 100 MHz clock stamps (s_memrealtime, one clock for the whole device) of the first / last eight
workgroups of the pre-pass and of the first eight of the GEMM, taken from the LAST step of a loop of steps.

    FQL_INT4_LIB=tools/micro/libfql_trace.so python tools/trace_step.py [--seconds 2] [--counts 128,128,...]
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--counts", default="", help="rows per expert, comma separated (default: 128 each)")
ap.add_argument("--precision", default="default")
ap.add_argument("--one-launch", action="store_true", help="the FUSED form: pre-pass as the GEMM kernel's first phase")
a = ap.parse_args()
dev = torch.device("cuda:0")
E, K, N = 8, 4096, 11008
counts = [int(c) for c in a.counts.split(",")] if a.counts else [128] * E
T = sum(counts)
g = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(4):
    q = [fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02) for _ in range(E)]
    sets.append(tuple(torch.stack([t[i] for t in q]) for i in range(3)))
x = torch.randn(T, K, device=dev, generator=g)
tpe = torch.tensor(counts, dtype=torch.int32, device=dev)
offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
lib = _native.lib()
if a.one_launch:
    lib.fql_tune_set_fused(1)
t0 = time.time(); i = 0; n_timed = 0; timing = False
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
while time.time() - t0 < a.seconds:
    if time.time() - t0 > a.seconds * 0.5 and not timing:
        ev0.record(); timing = True
    for _ in range(20):
        P, S, Z = sets[i % 4]; i += 1
        y = ops.moe_forward(P, S, Z, x, None, tpe, offs, precision=a.precision)
        n_timed += 1 if timing else 0
ev1.record(); torch.cuda.synchronize()
print(f"step wall (HIP events over the second half, {n_timed} steps): {ev0.elapsed_time(ev1) / max(n_timed, 1) * 1e3:.1f} us")


def fetch(name, n):
    buf = (ctypes.c_ulonglong * n)()
    fn = getattr(lib, name)
    fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
    assert fn(buf) == 0
    return np.array(buf[:], dtype=np.int64)


act = fetch("fql_debug_trace_act", 256).reshape(16, 16)
w4 = fetch("fql_debug_trace_w4", 512).reshape(8, 64)
if a.one_launch:
    act[:] = 0
    t00 = min(int(w4[b][56]) for b in range(8) if w4[b][56] > 0)
    print("one launch: GEMM kernel entry | row groups counted | tile table | own rows quantised + published | tiles' rows there | first stage parked")
    for b in range(8):
        if w4[b][56] > 0:
            print(f"  wg {b}: " + " | ".join(f"{(int(w4[b][i]) - t00) / 100.0:6.2f}" for i in (56, 57, 58, 60, 61, 59)))
else:
    t00 = min(int(v) for v in act[:, 0] if v > 0)
us = lambda v: (int(v) - t00) / 100.0 if v > 0 else float("nan")
print("pre-pass (us after its first workgroup's entry): entry | rows looked up | loads in, row max known | quantised, stores issued | row sums written | exit")
for s in range(16):
    if act[s][0] > 0:
        print(f"  act wg slot {s:2d}: " + " | ".join(f"{us(act[s][i]):6.2f}" for i in range(6)))
KT = K // 256
per = 2 + KT + 2 + 1
print("GEMM (same clock): entry | row groups counted | tile table | first stage parked || per visit: start, K loop done (shader cycles from start), end")
for b in range(8):
    row = w4[b]
    if row[56] <= 0:
        continue
    vis = [int(v) for v in row[:42] if v > 0]
    txt = []
    for k in range(0, len(vis) - per + 1, per):
        c0 = vis[k + 1]
        txt.append(f"start {us(vis[k]):6.2f} kloop {vis[k + 2 + KT] - c0} epi {vis[k + 3 + KT] - vis[k + 2 + KT]} end {us(vis[k + per - 1]):6.2f}")
    print(f"  gemm wg {b}: " + " | ".join(f"{us(row[i]):6.2f}" for i in (56, 57, 58, 59)) + " || " + " ; ".join(txt))
