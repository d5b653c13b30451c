#!/usr/bin/env python3
"""Diagnostic (FQL_STAMP build): in-kernel cycles / real time of the K loop and the epilogue of one workgroup."""
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
limbs, delta, rowsum = ops.act_quant(x, precision=prec, tokens_per_expert=tpe, input_offsets=offs)
out = torch.empty((x.shape[0], N), device=dev)
for cfg in [int(c) for c in sys.argv[1].split(",")]:
    for _ in range(20):
        rc = tune(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), P.data_ptr(), S.data_ptr(), Z.data_ptr(),
                  tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, x.shape[0], K, N, prec, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    assert rd(buf) == 0
    print(f"cfg {cfg}: workgroup 8, per wave: K-loop cycles / us (100 MHz realtime), epilogue cycles / us")
    for w in range(8):
        v = [buf[w * 8 + i] for i in range(4)]
        clk = v[0] / (v[1] / 100.0) / 1e3 if v[1] else 0
        print(f"  wave {w}: loop {v[0]:8d} cyc {v[1]/100.0:7.2f} us (clk {clk:.2f} GHz)   epilogue {v[2]:7d} cyc {v[3]/100.0:6.2f} us")
