#!/usr/bin/env python3
"""Is the headline GEMM loop clock- (power-) limited?  In-kernel clock of gemm_i8_kernel<3,4,2,3,2,1> (configs[2]) under
sustained load, from a -DFQL_TRACE build: wave 0 of eight workgroups stamps s_memtime (shader clock) at every 256-k
stage and s_memrealtime (100 MHz) at the tile boundaries (MI355X_MICROARCH.md 'DVFS give-back' item 6).

    FQL_INT4_LIB=tools/micro/libfql_trace.so python tools/power_evidence.py [--zero-acts] [--zero-weights] [--seconds 2]

Prints the wall time and shader clock of a tile and the cycles / wall time of a K stage; with an -DFQL_ABLATE=1 build
(no MFMA issued, everything else in place) the same numbers for the non-matrix skeleton."""
import argparse, ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native, routing as R

ap = argparse.ArgumentParser()
ap.add_argument("--zero-acts", action="store_true")
ap.add_argument("--zero-weights", action="store_true")
ap.add_argument("--seconds", type=float, default=2.0)
ap.add_argument("--cfg", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
E, K, N, T = 8, 4096, 11008, 1024
g = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(4):
    P, S, Z = [], [], []
    for e in range(E):
        w = torch.randn(N, K, device=dev, generator=g) * 0.02
        p, s, z = fq.quantize_weights(w)
        if a.zero_weights:
            p.zero_()
        P.append(p); S.append(s); Z.append(z)
    sets.append((torch.stack(P), torch.stack(S), torch.stack(Z)))
x = torch.randn(T, K, device=dev, generator=g)
if a.zero_acts:
    x.zero_()
tpe = torch.full((E,), T // E, dtype=torch.int32, device=dev)
offs = (torch.arange(E, device=dev, dtype=torch.int32) * (T // E))
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
print(f"kernel wall (HIP events over the second half, {n_timed} launches): {ev0.elapsed_time(ev1) / max(n_timed, 1) * 1e3:.1f} us")
buf = (ctypes.c_ulonglong * 512)()
fn = lib.fql_debug_trace_wide
fn.argtypes = [ctypes.c_void_p]; fn.restype = ctypes.c_int
assert fn(buf) == 0
arr = np.array(buf[:], dtype=np.int64).reshape(8, 64)
walls, clocks, stage_cyc = [], [], []
for b in range(8):
    row = arr[b]; row = row[row > 0]
    # layout per tile: real0, clk0, stage starts (KT = 16), k-loop done, epilogue issued, real1
    per = 2 + 16 + 2 + 1
    k = 0
    while k + per <= len(row):
        r0, c0 = int(row[k]), int(row[k + 1])
        st = [int(v) - c0 for v in row[k + 2:k + 2 + 16]]
        kd = int(row[k + 18]) - c0; ep = int(row[k + 19]) - c0; r1 = int(row[k + 20])
        wall = (r1 - r0) / 100.0
        if 5 < wall < 500 and ep > 0:
            walls.append(wall); clocks.append(ep / wall / 1000.0)
            stage_cyc.append((st[-1] - st[1]) / 14.0)
        k += per
if walls:
    w, c, s = np.median(walls), np.median(clocks), np.median(stage_cyc)
    print(f"tiles sampled {len(walls)}: tile wall {w:.1f} us, shader clock {c:.3f} GHz (min {min(clocks):.3f} max {max(clocks):.3f}), "
          f"K stage {s:.0f} cycles = {s / c / 1000.0:.2f} us (pure MFMA issue: 4608 cycles)")
else:
    print("no complete tile records", arr[0][:24])
