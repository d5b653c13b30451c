"""Expert-parallel step on the GPU with TWO ranks sharing one card: the device-side routing kernels (plan,
regroup index, fused gathers, fused combine) and the fused grouped GEMM, with the three all-to-alls carried by
gloo through host memory (test only: on a real node the same calls go over RCCL / xGMI).  The 2-rank result must
equal the 1-GPU grouped computation bit for bit (integer-exact GEMM: a row's result does not depend on the rank
that computed it; top-2 combine has a single addition)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(E, K, N, tokens, top_k, seed):
    import fused_int4_amd as fq
    g = torch.Generator().manual_seed(seed)
    P, S, Z = [], [], []
    for _ in range(E):
        w = torch.randn(N, K, generator=g) * 0.05
        if N * K > (1 << 22):           # BASELINE shapes: the GPU quantiser (bit-exact with the host one), then back to the host
            p, s, z = (t.cpu() for t in fq.quantize_weights(w.cuda()))
        else:
            p, s, z = fq.quantize_weights(w)
        P.append(p); S.append(s); Z.append(z)
    x = torch.randn(tokens, K, generator=g)
    w, idx = torch.topk(torch.softmax(torch.randn(tokens, E, generator=g), -1), top_k, dim=-1)
    return torch.stack(P), torch.stack(S), torch.stack(Z), x, idx, w / w.sum(-1, keepdim=True)


def _worker(rank, world, port, E, K, N, tokens, top_k, seed, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import fused_int4_amd.ep as ep_mod
        from fused_int4_amd import ops, routing as R
        real = dist.all_to_all_single

        def through_host(output, input, output_split_sizes=None, input_split_sizes=None, group=None):
            o = torch.empty(output.shape, dtype=output.dtype)
            real(o, input.cpu().contiguous(), output_split_sizes, input_split_sizes, group=group)
            output.copy_(o)
        ep_mod.dist.all_to_all_single = through_host          # gloo has no device all-to-all
        dev = torch.device("cuda:0")
        P, S, Z, x, idx, w = _problem(E, K, N, tokens, top_k, seed)
        per = tokens // world
        sl = slice(rank * per, (rank + 1) * per)
        shard = ep_mod.ExpertParallelMoE.shard
        ep = ep_mod.ExpertParallelMoE(E, shard(P, rank, world).to(dev), shard(S, rank, world).to(dev),
                                      shard(Z, rank, world).to(dev))
        y = ep(x[sl].contiguous().to(dev), idx[sl].contiguous().to(dev), w[sl].contiguous().to(dev))
        # the 1-GPU computation of the same tokens
        xg, tpe, offs, inv = R.dispatch_grouped(x.to(dev), idx.to(dev), E)
        ref = R.combine_grouped(ops.moe_forward(P.to(dev), S.to(dev), Z.to(dev), xg, None, tpe, offs), w.to(dev), inv, top_k)
        ret[rank] = bool(torch.equal(y, ref[sl]))
        ret[world + rank] = float((y - ref[sl]).abs().max())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("E,K,N,tokens", [(4, 256, 136, 64), (2, 256, 136, 18),
                                          (8, 4096, 11008, 512)])     # BASELINE.json configs[3]'s shape, 2 of its 8 ranks' worth
def test_two_ranks_one_gpu_equal_single_gpu(E, K, N, tokens):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E, K, N, tokens, 2, 77, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert all(ret[r] for r in range(world)), dict(ret)
