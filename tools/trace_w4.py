#!/usr/bin/env python3
"""In-kernel timeline and clock of the one-wave-per-SIMD GEMM (csrc/fql_gemm_w4.h) at BASELINE configs[2] under sustained
load, from a -DFQL_TRACE build (make -C .../csrc trace): wave 0 of eight workgroups stamps s_memtime (shader clock) at
every 256-k stage and s_memrealtime (100 MHz) at the visit boundaries.

    FQL_INT4_LIB=tools/micro/libfql_trace.so python tools/trace_w4.py [--cfg 300] [--zero-acts] [--seconds 2]
"""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native

ap = argparse.ArgumentParser()
ap.add_argument("--zero-acts", action="store_true")
ap.add_argument("--abs-acts", action="store_true", help="|randn| activations: non-negative top limb")
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--cfg", type=int, default=300)
ap.add_argument("--counts", default="", help="rows per expert, comma separated (default: 128 each)")
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
if a.zero_acts:
    x.zero_()
if a.abs_acts:
    x = x.abs()
tpe = torch.tensor(counts, dtype=torch.int32, device=dev)
offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
limbs, delta, rowsum = ops.act_quant(x, precision="exact", tokens_per_expert=tpe, input_offsets=offs)
out = torch.empty((T, N), device=dev)
lib = _native.lib()
t0 = time.time(); i = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n_timed = 0
timing = False
while time.time() - t0 < a.seconds:
    if time.time() - t0 > a.seconds * 0.5 and not timing:
        ev0.record(); timing = True
    for _ in range(20):
        P, S, Z = sets[i % 4]; i += 1
        rc = ops.tune_gemm_i8(a.cfg, limbs, delta, rowsum, P, S, Z, tpe, offs, out, E, T, K, N, "exact")
        assert rc == 0
        n_timed += 1 if timing else 0
ev1.record(); torch.cuda.synchronize()
print(f"cfg {a.cfg}: kernel wall (HIP events over the second half, {n_timed} launches): {ev0.elapsed_time(ev1) / max(n_timed, 1) * 1e3:.1f} us")
name = "fql_debug_trace_w4" if a.cfg >= 300 else "fql_debug_trace_wide"
buf = (ctypes.c_ulonglong * 512)()
fn = getattr(lib, name)
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf) == 0
arr = np.array(buf[:], dtype=np.int64).reshape(8, 64)
KT = K // 256
per = 2 + KT + 2 + 1          # real0, clk0, KT stage starts, k-loop done, epilogue done, real1
for b in range(8):
    row = arr[b][:42]; row = row[row > 0]
    k = 0; v = 0
    while k + per <= len(row):
        r0, c0 = int(row[k]), int(row[k + 1])
        st = [int(x) - c0 for x in row[k + 2:k + 2 + KT]]
        kd = int(row[k + 2 + KT]) - c0; ep = int(row[k + 3 + KT]) - c0; r1 = int(row[k + 4 + KT])
        wall = (r1 - r0) / 100.0
        d = np.diff(st + [kd])
        print(f"wg {b} visit {v}: wall {wall:6.1f} us  clock {ep / wall / 1000.0:.3f} GHz  first stage at {st[0]}  stages {d.tolist()}  "
              f"epilogue {ep - kd} cycles")
        k += per; v += 1
if a.cfg >= 300:
    print('stamps per workgroup:', [int((arr[b][:56] > 0).sum()) for b in range(8)])
    # kernel-entry / prologue stamps (100 MHz): entry, row groups counted, tile table built, first stage parked; relative to
    # the earliest entry of the eight traced workgroups
    t00 = min(int(arr[b][56]) for b in range(8) if arr[b][56] > 0)
    for b in range(8):
        sp = [(int(arr[b][i]) - t00) / 100.0 for i in (56, 57, 58, 59)]
        vis = [int(x) for x in arr[b][:42] if x > 0]
        starts = [(vis[k] - t00) / 100.0 for k in range(0, len(vis) - per + 1, per)]
        ends = [(vis[k + per - 1] - t00) / 100.0 for k in range(0, len(vis) - per + 1, per)]
        fr = [int(x) for x in arr[b][42:48] if x > 0]
        if len(fr) > 1 and len(vis) >= per:
            kd0 = vis[2 + KT]; ep0 = vis[3 + KT]
            print(f"wg {b}: first visit's epilogue, cycles per fragment {np.diff([kd0] + fr + [ep0]).tolist()} (first entry: K loop end -> fragment 0)")
        print(f"wg {b}: entry {sp[0]:.2f} groups {sp[1]:.2f} table {sp[2]:.2f} primed {sp[3]:.2f} | visits start {starts} end {ends} (us after the first entry)")
