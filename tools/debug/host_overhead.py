import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import fused_int4_amd as fq
from fused_int4_amd import ops, _native, routing as R
dev = torch.device("cuda:0")
E, K, N, T = 8, 4096, 11008, 1024
g = torch.Generator(device=dev).manual_seed(0)
P, S, Z = [], [], []
for e in range(E):
    p, s, z = fq.quantize_weights(torch.randn(N, K, device=dev, generator=g) * 0.02)
    P.append(p); S.append(s); Z.append(z)
P, S, Z = torch.stack(P), torch.stack(S), torch.stack(Z)
x = torch.randn(T, K, device=dev, generator=g)
tpe = torch.full((E,), T // E, dtype=torch.int32, device=dev)
offs = (torch.arange(E, device=dev, dtype=torch.int32) * (T // E))
L = _native.lib()
def timeit(fn, n=200, name=""):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    th = time.perf_counter() - t0
    torch.cuda.synchronize(); tt = time.perf_counter() - t0
    print(f"{name:50s} host-enqueue {th/n*1e6:8.1f} us/call   total {tt/n*1e6:8.1f} us/call")
timeit(lambda: ops.moe_forward(P, S, Z, x, None, tpe, offs), name="ops.moe_forward (product call)")
timeit(lambda: ops.moe_forward(P, S, Z, x, None, tpe, offs, precision="int8"), name="ops.moe_forward int8")
ws = torch.empty(L.fql_moe_workspace_bytes(E, T, K, N, 3), dtype=torch.uint8, device=dev)
out = torch.empty((T, N), device=dev)
st = torch.cuda.current_stream().cuda_stream
def raw():
    rc = L.fql_moe_fwd_f32(P.data_ptr(), S.data_ptr(), Z.data_ptr(), x.data_ptr(), tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, T, K, N, 3, ws.data_ptr(), ws.numel(), st)
    assert rc == 0
timeit(raw, name="fql_moe_fwd_f32 with preallocated buffers")
timeit(lambda: torch.empty(L.fql_moe_workspace_bytes(E, T, K, N, 3), dtype=torch.uint8, device=dev), name="torch.empty(workspace)")
timeit(lambda: torch.empty((T, N), device=dev), name="torch.empty(out)")
print("workspace bytes", L.fql_moe_workspace_bytes(E, T, K, N, 3))
limbs, delta, rowsum = ops.act_quant(x, precision="exact", tokens_per_expert=tpe, input_offsets=offs)
timeit(lambda: ops.act_quant(x, precision="exact", tokens_per_expert=tpe, input_offsets=offs, out=(limbs, delta, rowsum)), name="act_quant only")
timeit(lambda: ops.gemm_i8(limbs, delta, rowsum, P, S, Z, tpe, offs, precision="exact", out=out), name="gemm_i8 only")
