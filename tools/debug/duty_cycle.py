#!/usr/bin/env python3
"""Does the headline GEMM run faster when the pre-pass sits between two launches of it?  Three phases of ~100 ms each in
ONE process, no idle gap between them: [pre-pass, GEMM] pairs, GEMM alone back to back, pairs again.  Run under
`rocprofv3 --kernel-trace` and feed the trace to this script with --analyse <kernel_trace.csv>."""
import csv
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    rows = [r for r in csv.DictReader(open(sys.argv[2])) if "gemm_i8_kernel" in r["Kernel_Name"] or "act_fused" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    seq = [("G" if "gemm" in r["Kernel_Name"] else "P", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r["Start_Timestamp"])) for r in rows]
    # GEMM launches classified by what preceded them
    after_p = [d for i, (k, d, _) in enumerate(seq) if k == "G" and i > 0 and seq[i - 1][0] == "P"]
    after_g = [d for i, (k, d, _) in enumerate(seq) if k == "G" and i > 0 and seq[i - 1][0] == "G"]
    def stat(name, v, lo, hi):
        v = v[lo:hi]
        v2 = sorted(v)
        print(f"{name}: n={len(v)} mean {sum(v) / len(v):.1f} us  median {v2[len(v2) // 2]:.1f}  min {v2[0]:.1f}  max {v2[-1]:.1f}")
    n1 = len(after_p) // 2
    stat("GEMM after a pre-pass, phase 1 (last 500)", after_p, n1 - 500, n1)
    stat("GEMM after a GEMM,     phase 2 (first 100)", after_g, 0, 100)
    stat("GEMM after a GEMM,     phase 2 (last 500) ", after_g, len(after_g) - 500, len(after_g))
    stat("GEMM after a pre-pass, phase 3 (first 100)", after_p, n1, n1 + 100)
    stat("GEMM after a pre-pass, phase 3 (last 500) ", after_p, len(after_p) - 500, len(after_p))
    t0 = seq[0][2]
    span = (seq[-1][2] - t0) / 1e6
    print(f"trace spans {span:.1f} ms, {len(seq)} launches")
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import fused_int4_amd as fq  # noqa: E402
from fused_int4_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
E, K, N, T = 8, 4096, 11008, 1024
g = torch.Generator(device=dev).manual_seed(1)
sets = []
for i in range(4):
    P, S, Z = [], [], []
    for e in range(E):
        p, s, z = fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02)
        P.append(p); S.append(s); Z.append(z)
    sets.append((torch.stack(P), torch.stack(S), torch.stack(Z)))
x = torch.randn(T, K, device=dev, generator=g)
tpe = torch.full((E,), T // E, dtype=torch.int32, device=dev)
offs = torch.arange(E, dtype=torch.int32, device=dev) * (T // E)
bufs = ops.act_quant(x, tokens_per_expert=tpe, input_offsets=offs)
out = torch.empty((T, N), device=dev)
NPH = 900


def pair(i):
    lm, dl, rs = ops.act_quant(x, tokens_per_expert=tpe, input_offsets=offs, out=bufs)
    P, S, Z = sets[i % 4]
    ops.gemm_i8(lm, dl, rs, P, S, Z, tpe, offs, out=out)


for i in range(NPH):
    pair(i)
lm, dl, rs = bufs
for i in range(NPH):
    P, S, Z = sets[i % 4]
    ops.gemm_i8(lm, dl, rs, P, S, Z, tpe, offs, out=out)
for i in range(NPH):
    pair(i)
torch.cuda.synchronize()
print("done")
