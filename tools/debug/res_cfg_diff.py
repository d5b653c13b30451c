"""Which tile configurations differ on a mixed heavy-tail problem, and where (debug aid)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from fused_int4_amd import ops, _native
lib = _native.lib()
rng = np.random.default_rng(99)
E, Nn, Kk = 5, 200, 768
counts = np.array([0, 7, 33, 70, 129], np.int32)
offs = (np.cumsum(counts) - counts).astype(np.int32)
T = int(counts.sum())
q = [O.quantize_weights((rng.standard_normal((Nn, Kk)) * 0.02).astype(np.float32)) for _ in range(E)]
P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
x = rng.standard_normal((T, Kk)).astype(np.float32)
for t in range(0, T, 5):
    x[t, rng.choice(Kk, 2, replace=False)] *= 800.0
x[40:56, :] = rng.standard_normal((16, Kk)).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dP, dS, dZ, dx, dc, do = d(P), d(S), d(Z), d(x), d(counts), d(offs)
limbs, delta, rowsum = ops.act_quant(dx, precision="exact", tokens_per_expert=dc, input_offsets=do)
flag = (delta[1] != 0).cpu().numpy()
cfgs = list(range(lib.fql_tune_num_configs())) + list(range(100, 100 + lib.fql_tune_num_rows32_configs())) + list(range(200, 200 + lib.fql_tune_num_rows16_configs()))
for rep in range(3):
    outs = {}
    for cfg in cfgs:
        out = torch.full((T, Nn), float("nan"), dtype=torch.float32, device="cuda")
        rc = ops.tune_gemm_i8(cfg, limbs, delta, rowsum, dP, dS, dZ, dc, do, out, E, T, Kk, Nn, "exact")
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy()
    bad = []
    for cfg, o in outs.items():
        diff = o != outs[0]
        if diff.any():
            rows = np.nonzero(diff.any(axis=1))[0]
            bad.append((cfg, int(diff.sum()), rows[:8].tolist(), flag[rows[:8]].tolist(), float(np.nanmax(np.abs(o - outs[0])))))
    print("rep", rep, "differing configs:", bad)
