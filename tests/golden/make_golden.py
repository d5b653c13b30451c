#!/usr/bin/env python3
"""Generate golden input/output vectors by RUNNING the reference's own Python on CPU.

Runs only in the build container (needs /root/reference, which does not exist on the
GPU box).  Nothing from the reference is copied: the outputs below are *data* --
inputs and the values the reference's functions returned for them -- stored as
``.npz`` (numpy, allow_pickle=False) next to this script.

Fixtures, reference entry points exercised:
  python/quantize.py:38-124    quantize_weights
  python/quantize.py:127-173   dequantize_weights
  python/quantize.py:176-202   reference_quantized_linear   (THE ORACLE)
  python/module.py:33-138      QuantizedLinear.from_linear / forward (CPU branch)
  python/moe_int4_module.py:19-80   quantize_weights_moe (per-tensor)
  benchmark/moe_grouped_gemm/moe_int4_module.py:21-130  QuantizedMoE
  benchmark/moe_grouped_gemm/routing.py:26-189  simulate_routing / dispatch / combine

Usage:  python tests/golden/make_golden.py
"""
import io
import contextlib
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    sys.path.insert(0, REF)
    from python.quantize import quantize_weights, dequantize_weights, reference_quantized_linear
    from python.module import QuantizedLinear
    with contextlib.redirect_stdout(io.StringIO()):      # import prints a "not available" warning
        import python.moe_int4_module as ref_moe
    from benchmark.moe_grouped_gemm.moe_int4_module import QuantizedMoE
    from benchmark.moe_grouped_gemm.routing import (
        simulate_routing, create_expert_inputs, combine_expert_outputs)

    torch.set_num_threads(1)          # fixed summation order inside sgemm for the stored outputs
    n = lambda t: t.detach().cpu().numpy()

    def save(name, **arrs):
        path = os.path.join(OUT, name + ".npz")
        np.savez(path, **arrs)
        print(f"{name}.npz  {os.path.getsize(path)/1024:.1f} KiB")

    # F1  tests/test_correctness.py:49-57  (16x32, seed 42) quantize + dequantize
    torch.manual_seed(42)
    w = torch.randn(16, 32)
    p, s, z = quantize_weights(w)
    save("f1_quant_16x32", weight=n(w), packed=n(p), scales=n(s), zero_points=n(z),
         dequant=n(dequantize_weights(p, s, z)))

    # F2  test_correctness.py:109-128,201-219  (64x128, seed 42, x[128])
    torch.manual_seed(42)
    w = torch.randn(64, 128)
    x = torch.randn(128)
    p, s, z = quantize_weights(w)
    save("f2_linear_64x128", weight=n(w), x=n(x), packed=n(p), scales=n(s), zero_points=n(z),
         out=n(reference_quantized_linear(x, p, s, z)))

    # F3  test_correctness.py:221-234  (256x512, seed 42, x[4,512])
    torch.manual_seed(42)
    w = torch.randn(256, 512)
    x = torch.randn(4, 512)
    p, s, z = quantize_weights(w)
    save("f3_linear_256x512_b4", x=n(x), packed=n(p), scales=n(s), zero_points=n(z),
         out=n(reference_quantized_linear(x, p, s, z)))

    # F4  test_correctness.py:93-103  constant rows (all 3.0) + other constant-row corner cases
    w = torch.ones(4, 8) * 3.0
    p, s, z = quantize_weights(w)
    w2 = torch.tensor([[0.0] * 8, [-2.5] * 8, [0.25] * 8, [1e-9] * 8, [-1e-9, 1e-9] * 4,
                       [7.0, -7.0] * 4], dtype=torch.float32)
    p2, s2, z2 = quantize_weights(w2)
    save("f4_constant_rows", weight=n(w), packed=n(p), scales=n(s), zero_points=n(z),
         dequant=n(dequantize_weights(p, s, z)),
         weight2=n(w2), packed2=n(p2), scales2=n(s2), zero_points2=n(z2),
         dequant2=n(dequantize_weights(p2, s2, z2)))

    # F5  test_correctness.py:236-253  (4096x4096, seed 7, x[4096]); keep a 64-row slice
    torch.manual_seed(7)
    w = torch.randn(4096, 4096)
    x = torch.randn(4096)
    p, s, z = quantize_weights(w)
    out = reference_quantized_linear(x, p, s, z)
    rows = np.arange(0, 4096, 64)
    save("f5_linear_4096_rows64", x=n(x), rows=rows, packed=n(p)[rows],
         scales=n(s)[rows], zero_points=n(z)[rows], out=n(out)[rows])

    # F6  tests/test_benchmark.py:33-53  QuantizedLinear.from_linear(nn.Linear(128,64)) seed 42
    torch.manual_seed(42)
    lin = torch.nn.Linear(128, 64, bias=False)
    ql = QuantizedLinear.from_linear(lin)
    x1 = torch.randn(128)
    torch.manual_seed(42)
    _ = torch.nn.Linear(128, 64, bias=False)
    x4 = torch.randn(4, 128)
    sd = ql.state_dict()
    save("f6_module_128x64", weight=n(lin.weight.data), x1=n(x1), x4=n(x4),
         packed_weights=n(sd["packed_weights"]), scales=n(sd["scales"]),
         zero_points=n(sd["zero_points"]), out1=n(ql(x1)), out4=n(ql(x4)),
         state_dict_keys=np.array(sorted(sd.keys())), extra_repr=np.array(ql.extra_repr()))

    # F7  python/moe_int4_module.py:19-80  quantize_weights_moe on 2 x (32x64) fp16 randn*0.02 seed 0
    torch.manual_seed(0)
    ws = [(torch.randn(32, 64) * 0.02).half() for _ in range(2)]
    pk, sc, zp = ref_moe.quantize_weights_moe(ws)
    # 3 experts, wider value range, seed 1 (zero-point away from the middle)
    torch.manual_seed(1)
    ws2 = [(torch.randn(16, 32) * (0.5 + e) + 0.3 * e).half() for e in range(3)]
    pk2, sc2, zp2 = ref_moe.quantize_weights_moe(ws2)
    moe_mod = ref_moe.MoEINT4.from_weights(ws)
    save("f7_moe_per_tensor", weights=np.stack([n(w) for w in ws]), packed=n(pk), scales=n(sc),
         zero_points=n(zp), weights2=np.stack([n(w) for w in ws2]), packed2=n(pk2), scales2=n(sc2),
         zero_points2=n(zp2), state_dict_keys=np.array(sorted(moe_mod.state_dict().keys())))

    # F8  QuantizedMoE (benchmark/moe_grouped_gemm/moe_int4_module.py) 4 experts 128->256,
    #     m_sizes from simulate_routing(64 tokens, 4 experts, top-2, 'skewed', seed 42) with one
    #     expert forced empty; fp32 and fp16 inputs (fp16 output-dtype quirk at :65-68).
    E, K, N = 4, 128, 256
    torch.manual_seed(42)
    wl = [(torch.randn(N, K) * 0.02).half() for _ in range(E)]
    qm = QuantizedMoE.from_fp16_weights(wl)
    routing = simulate_routing(64, E, 2, "skewed", device="cpu", seed=42)
    m_sizes = list(routing.tokens_per_expert)
    m_sizes[3] = 0
    torch.manual_seed(43)
    xin32 = [torch.randn(m, K) for m in m_sizes]
    xin16 = [t.half() for t in xin32]
    o32 = qm(xin32)
    o16 = qm(xin16)
    sd = qm.state_dict()
    arrs = dict(weights=np.stack([n(w) for w in wl]), m_sizes=np.array(m_sizes),
                x32=np.concatenate([n(t) for t in xin32]),
                out32=np.concatenate([n(t) for t in o32]),
                out16=np.concatenate([n(t) for t in o16]),
                empty_out_dtype=np.array(str(o32[3].dtype)),
                total_memory_bytes=np.array(qm.total_memory_bytes),
                state_dict_keys=np.array(sorted(sd.keys())))
    for e in range(E):
        arrs[f"packed{e}"] = n(sd[f"experts.{e}.packed_weights"])
        arrs[f"scales{e}"] = n(sd[f"experts.{e}.scales"])
        arrs[f"zero_points{e}"] = n(sd[f"experts.{e}.zero_points"])
    save("f8_quantized_moe", **arrs)

    # F9  routing.py  simulate_routing / create_expert_inputs / combine_expert_outputs round trip
    T, E, top_k, K = 16, 4, 2, 8
    r = simulate_routing(T, E, top_k, "skewed", device="cpu", seed=42)
    torch.manual_seed(5)
    x = torch.randn(T, K)
    ein, perm = create_expert_inputs(x, r, E, top_k)
    eout = [t * (e + 1.0) for e, t in enumerate(ein)]          # identity-like expert: scale by e+1
    comb = combine_expert_outputs(eout, r, perm, top_k)
    rr = simulate_routing(32, 8, 2, "random", device="cpu", seed=7)
    save("f9_routing", x=n(x), expert_indices=n(r.expert_indices), expert_weights=n(r.expert_weights),
         tokens_per_expert=np.array(r.tokens_per_expert),
         expert_token_offsets=np.array(r.expert_token_offsets),
         expert_inputs=np.concatenate([n(t) for t in ein]), permutation=n(perm), combined=n(comb),
         random_indices=n(rr.expert_indices), random_weights=n(rr.expert_weights),
         random_tokens_per_expert=np.array(rr.tokens_per_expert))


if __name__ == "__main__":
    main()
