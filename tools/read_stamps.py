#!/usr/bin/env python3
"""Diagnostic: run one GEMM of each requested cfg with the FQL_STAMP build and print per-wave phase cycles."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native, routing as R
dev = torch.device("cuda:0")
lib = _native.lib()
tune = lib.fql_tune_gemm_i8_f32
tune.restype = ctypes.c_int
tune.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p]
rd = lib.fql_tune_read_stamps
E, K, N = 8, 4096, 11008
g = torch.Generator(device=dev).manual_seed(1)
P, S, Z = [], [], []
for e in range(E):
    p, s, z = fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02)
    P.append(p); S.append(s); Z.append(z)
P, S, Z = torch.stack(P), torch.stack(S), torch.stack(Z)
route = R.balanced_routing(512, E, 2, device=dev, seed=42)
x, tpe, offs, _ = R.dispatch_grouped(torch.randn(512, K, device=dev, generator=g), route.expert_indices, E)
x = x.contiguous()
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 3
limbs, delta, rowsum = ops.act_quant(x, precision=prec)
out = torch.empty((x.shape[0], N), device=dev)
for cfg in [int(c) for c in sys.argv[1].split(",")]:
    for _ in range(3):
        rc = tune(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), P.data_ptr(), S.data_ptr(), Z.data_ptr(),
                  tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, x.shape[0], K, N, prec, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert rd(buf) == 0
    print(f"cfg {cfg}: per wave [R, bar1, M, bar2] cycles summed over the K loop (128 phases pairs)")
    for w in range(8):
        v = [buf[w * 8 + i] for i in range(4)]
        print(f"  wave {w}: R={v[0]:8d} bar1={v[1]:8d} M={v[2]:8d} bar2={v[3]:8d} total={sum(v):8d}  per-step R={v[0]/128:.0f} b1={v[1]/128:.0f} M={v[2]/128:.0f} b2={v[3]/128:.0f}")
