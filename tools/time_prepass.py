import os, sys, torch
sys.path.insert(0, "/root/repo")
import fused_int4_amd as fq
from fused_int4_amd import ops
dev = torch.device("cuda:0")
if len(sys.argv) > 1:
    from fused_int4_amd import _native
    print("act single rows", _native.lib().fql_tune_set_act_single_rows(int(sys.argv[1])), "->", sys.argv[1])
E, T, K = 8, 1024, 4096
x = torch.randn(T, K, device=dev)
xs = [torch.randn(T, K, device=dev) for _ in range(24)]   # rotate: 400 MB > infinity cache
tpe = torch.full((E,), T // E, dtype=torch.int32, device=dev); offs = (torch.arange(E, device=dev, dtype=torch.int32) * (T // E))
bufs = ops.act_quant(x, tokens_per_expert=tpe, input_offsets=offs)
for xx in xs[:4]: ops.act_quant(xx, tokens_per_expert=tpe, input_offsets=offs, out=bufs)
torch.cuda.synchronize()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(96)]
for i, (a, b) in enumerate(ev):
    a.record(); ops.act_quant(xs[i % 24], tokens_per_expert=tpe, input_offsets=offs, out=bufs); b.record()
torch.cuda.synchronize()
t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
print(f"pre-pass (events around the call): median {t[48]:.1f} us  min {t[0]:.1f} us")
