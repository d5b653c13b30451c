#!/usr/bin/env python3
"""Debug: per-phase s_memtime stamps of wave 0 of the first workgroups of one rows32 launch
(needs a library built with -DFQL_TRACE: FQL_INT4_LIB=tools/micro/libfql_trace.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
m = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0"); E, K, N = (int(sys.argv[3]) if len(sys.argv) > 3 else 8), 4096, 11008
g = torch.Generator(device=dev).manual_seed(0)
sets = []
for _ in range(3):
    P, S, Z = [], [], []
    for e in range(E):
        p, s, z = fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02)
        P.append(p); S.append(s); Z.append(z)
    sets.append((torch.stack(P), torch.stack(S), torch.stack(Z)))
T = m * E
x = torch.randn(T, K, device=dev, generator=g)
tpe = torch.full((E,), m, dtype=torch.int32, device=dev); offs = (torch.arange(E, device=dev, dtype=torch.int32) * m)
limbs, delta, rowsum = ops.act_quant(x, precision="exact", tokens_per_expert=tpe, input_offsets=offs)
lib = _native.lib()
tune = lib.fql_tune_gemm_i8_f32; tune.restype = ctypes.c_int
tune.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
out = torch.empty((T, N), device=dev)
st = torch.cuda.current_stream().cuda_stream
for i in range(3):
    P, S, Z = sets[i]
    rc = tune(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), P.data_ptr(), S.data_ptr(), Z.data_ptr(),
              tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, T, K, N, 3, st, None, 0)
    assert rc == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
fn = lib.fql_debug_trace if cfg >= 100 else lib.fql_debug_trace_wide
fn.argtypes = [ctypes.c_void_p]
assert fn(buf) == 0
a = np.array(buf[:], dtype=np.int64).reshape(8, 64)
for b in range(4):
    row = a[b]; row = row[row > 0]
    if cfg >= 100:
        print(f"block {b} (x100 shader cycles): " + " ".join(f"{(v - row[0]) / 100.0:.2f}" for v in row))
    else:
        # per tile: [real0, clk0, stage starts..., kdone, epi, real1]
        print(f"block {b} raw deltas:")
        i = 0
        while i + 2 < len(row):
            r0, c0 = row[i], row[i + 1]
            j = i + 2
            seq = []
            while j < len(row) and abs(int(row[j]) - int(c0)) < 10**9 and len(seq) < 19:
                seq.append(int(row[j]) - int(c0)); j += 1
            # the last element of seq is actually real1 if it is far from c0; handle by layout: stages(KT) + kdone + epi then real1
            print("   shader-clock cycles since tile start:", seq)
            if j < len(row):
                r1 = row[j]
                dt_us = (int(r1) - int(r0)) / 100.0
                if seq:
                    print(f"   tile wall {dt_us:.2f} us (100 MHz counter) -> shader clock {seq[-1] / dt_us / 1000:.3f} GHz")
            i = j + 1
