"""Which fp8 tile configurations differ from each other, and where (debug aid)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O
from fused_int4_amd import ops, _native
lib = _native.lib()
tune = lib.fql_tune_gemm_i8_f32
tune.restype = ctypes.c_int
tune.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
rng = np.random.default_rng(123)
E, N, K = 5, 200, 768
counts = np.array([0, 7, 33, 70, 129], np.int32)
offs = (np.cumsum(counts) - counts).astype(np.int32)
T = int(counts.sum())
q = [O.quantize_weights((rng.standard_normal((N, K)) * 0.02).astype(np.float32)) for _ in range(E)]
P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
x = rng.standard_normal((T, K)).astype(np.float32)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
dP, dS, dZ, dx, dc, do = d(P), d(S), d(Z), d(x), d(counts), d(offs)
limbs, delta, rowsum = ops.act_quant(dx, precision="fp8", tokens_per_expert=dc, input_offsets=do)
st = torch.cuda.current_stream().cuda_stream
outs = {}
for rep in range(2):
    for cfg in (1, 5, 6, 7, 8, 11, 12):
        out = torch.full((T, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = tune(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), dP.data_ptr(), dS.data_ptr(), dZ.data_ptr(),
                  dc.data_ptr(), do.data_ptr(), out.data_ptr(), E, T, K, N, 8, st, None, 0)
        torch.cuda.synchronize()
        o = out.cpu().numpy()
        if rep == 1:
            print("cfg", cfg, "repeatable", np.array_equal(o, outs[cfg]))
        outs[cfg] = o
ref = outs[1]
for cfg, o in outs.items():
    diff = o != ref
    rows = np.nonzero(diff.any(axis=1))[0]
    cols = np.nonzero(diff.any(axis=0))[0]
    print(f"cfg {cfg}: {diff.sum()} of {diff.size} differ; rows {rows[:12]}.. ({len(rows)}), cols {cols[:12]}.. ({len(cols)}); max rel {np.nanmax(np.abs(o-ref)/(np.abs(ref)+1e-30)):.2e}")
# position inside the 32-row blocks of the differing rows
for cfg in (5, 11, 12):
    diff = outs[cfg] != ref
    print(cfg, "differing rows mod 32 histogram:", np.bincount(np.nonzero(diff.any(axis=1))[0] % 32, minlength=32))
