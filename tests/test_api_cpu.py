"""CPU tests of the host-side mirror of the reference API (no GPU needed).

Restates the reference's own CPU test-suite (tests/test_correctness.py:39-168 and
tests/test_benchmark.py:23-76: 21 CPU test items) against this package, and pins the package's
quantisers / routing bit-exactly against the golden vectors generated from the reference."""
import numpy as np
import pytest
import torch

import fused_int4_amd as fq
from fused_int4_amd import quantize_weights, dequantize_weights, reference_quantized_linear, QuantizedLinear
from conftest import load_golden
from oracle import oracle as O


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


# ---- reference tests/test_correctness.py:39-103  TestQuantizeRoundTrip
@pytest.mark.parametrize("shape,seed", [((16, 32), 42), ((256, 512), 123), ((1024, 1024), 7)])
def test_round_trip(shape, seed):
    torch.manual_seed(seed)
    w = torch.randn(*shape)
    packed, scales, zp = quantize_weights(w)
    rec = dequantize_weights(packed, scales, zp)
    assert rec.shape == w.shape
    assert torch.allclose(w, rec, atol=0.5)


def test_packing_shape_and_range():
    packed, scales, zp = quantize_weights(torch.randn(64, 128))
    assert packed.shape == (64, 64) and scales.shape == (64,) and zp.shape == (64,)
    assert packed.dtype == torch.uint8 and scales.dtype == torch.float32 and zp.dtype == torch.float32
    assert packed.max() <= 255 and packed.min() >= 0
    assert torch.all(zp == torch.round(zp)) and zp.min() >= 0 and zp.max() <= 15


def test_constant_row():
    w = torch.ones(4, 8) * 3.0
    packed, scales, zp = quantize_weights(w)
    rec = dequantize_weights(packed, scales, zp)
    assert not torch.isnan(rec).any()
    assert torch.allclose(w, rec, atol=0.5)


def test_quantize_asserts():
    with pytest.raises(AssertionError):
        quantize_weights(torch.randn(8))
    with pytest.raises(AssertionError):
        quantize_weights(torch.randn(4, 7))


# ---- reference tests/test_correctness.py:106-168  TestReferenceLinear
def test_matches_f_linear():
    torch.manual_seed(42)
    w = torch.randn(64, 128)
    x = torch.randn(128)
    packed, scales, zp = quantize_weights(w)
    out_ref = reference_quantized_linear(x, packed, scales, zp)
    out_manual = torch.nn.functional.linear(x, dequantize_weights(packed, scales, zp))
    assert torch.allclose(out_ref, out_manual, atol=1e-5)
    assert reference_quantized_linear(torch.randn(8, 128), packed, scales, zp).shape == (8, 64)


def test_accuracy_vs_fp32():
    torch.manual_seed(42)
    w = torch.randn(256, 512)
    x = torch.randn(512)
    out_fp32 = torch.nn.functional.linear(x, w)
    out_q = reference_quantized_linear(x, *quantize_weights(w))
    assert (out_fp32 - out_q).abs().mean() < 3.0
    assert torch.nn.functional.cosine_similarity(out_fp32[None], out_q[None]) > 0.95


# ---- reference tests/test_benchmark.py:23-76  TestBenchmarkSmoke (CPU branch of the module)
SIZES = [(128, 64), (512, 256), (1024, 1024), (2048, 2048)]


@pytest.mark.parametrize("in_dim,out_dim", SIZES)
def test_module_forward_cpu(in_dim, out_dim):
    torch.manual_seed(42)
    ql = QuantizedLinear.from_linear(torch.nn.Linear(in_dim, out_dim, bias=False))
    out = ql(torch.randn(in_dim))
    assert out.shape == (out_dim,) and not torch.isnan(out).any()
    out = ql(torch.randn(4, in_dim))
    assert out.shape == (4, out_dim) and not torch.isnan(out).any()
    fp32_bytes = in_dim * out_dim * 4
    assert fp32_bytes / ql.packed_weights.nelement() >= 7.5


# ---- golden vectors: bit-exact quantisers, module surface
@pytest.mark.parametrize("name", ["f1_quant_16x32", "f2_linear_64x128"])
def test_quantize_bit_exact_vs_reference(name):
    g = load_golden(name)
    p, s, z = quantize_weights(t(g["weight"]))
    assert np.array_equal(p.numpy(), g["packed"])
    assert np.array_equal(s.numpy(), g["scales"])
    assert np.array_equal(z.numpy(), g["zero_points"])
    if "dequant" in g:
        assert np.array_equal(dequantize_weights(p, s, z).numpy(), g["dequant"])


def test_constant_row_corner_cases_bit_exact():
    g = load_golden("f4_constant_rows")
    for sfx in ("", "2"):
        p, s, z = quantize_weights(t(g["weight" + sfx]))
        assert np.array_equal(p.numpy(), g["packed" + sfx])
        assert np.array_equal(s.numpy(), g["scales" + sfx])
        assert np.array_equal(z.numpy(), g["zero_points" + sfx])
        assert np.array_equal(dequantize_weights(p, s, z).numpy(), g["dequant" + sfx])


def test_module_surface_matches_reference():
    g = load_golden("f6_module_128x64")
    lin = torch.nn.Linear(128, 64, bias=False)
    lin.weight.data = t(g["weight"])
    ql = QuantizedLinear.from_linear(lin)
    sd = ql.state_dict()
    assert sorted(sd.keys()) == list(g["state_dict_keys"]) == ["packed_weights", "scales", "zero_points"]
    assert np.array_equal(sd["packed_weights"].numpy(), g["packed_weights"])
    assert np.array_equal(sd["scales"].numpy(), g["scales"]) and np.array_equal(sd["zero_points"].numpy(), g["zero_points"])
    assert ql.extra_repr() == str(g["extra_repr"]) == "in_features=128, out_features=64, bits=4"
    assert torch.allclose(ql(t(g["x1"])), t(g["out1"]), atol=1e-5)
    assert torch.allclose(ql(t(g["x4"])), t(g["out4"]), atol=1e-5)
    # state_dict round trip into a freshly constructed module
    ql2 = QuantizedLinear(128, 64)
    ql2.load_state_dict(sd)
    assert torch.equal(ql2(t(g["x1"])), ql(t(g["x1"])))
    # the reference asserts `bias is None` (python/module.py:84); here a bias is carried as one more float32 buffer
    #  and the bias-free module keeps exactly the reference's state_dict
    lb = torch.nn.Linear(8, 4, bias=True)
    qb = QuantizedLinear.from_linear(lb)
    assert sorted(qb.state_dict().keys()) == ["bias", "packed_weights", "scales", "zero_points"]
    xb = torch.randn(3, 8)
    assert torch.equal(qb(xb), fq.reference_quantized_linear(xb, qb.packed_weights, qb.scales, qb.zero_points) + lb.bias.data)
    qb2 = QuantizedLinear(8, 4, bias=True)
    qb2.load_state_dict(qb.state_dict())
    assert torch.equal(qb2(xb), qb(xb))
    assert set(fq.__all__) >= {"quantize_weights", "dequantize_weights", "reference_quantized_linear", "QuantizedLinear"}


def test_quantize_weights_moe_bit_exact():
    g = load_golden("f7_moe_per_tensor")
    for sfx in ("", "2"):
        p, s, z = fq.quantize_weights_moe([t(w) for w in g["weights" + sfx]])
        assert np.array_equal(p.numpy(), g["packed" + sfx])
        assert np.array_equal(s.numpy(), g["scales" + sfx])
        assert np.array_equal(z.numpy(), g["zero_points" + sfx])
    mod = fq.MoEINT4.from_weights([t(w) for w in g["weights"]])
    assert sorted(mod.state_dict().keys()) == list(g["state_dict_keys"])
    assert mod.packed_weights.shape == (2, 32, 32) and mod.scales.shape == (2, 32)


def test_quantized_moe_cpu_path_and_quirks():
    g = load_golden("f8_quantized_moe")
    moe = fq.QuantizedMoE.from_fp16_weights([t(w) for w in g["weights"]])
    for e in range(4):
        assert np.array_equal(moe.experts[e].packed_weights.numpy(), g[f"packed{e}"])
        assert np.array_equal(moe.experts[e].scales.numpy(), g[f"scales{e}"])
    assert moe.total_memory_bytes == int(g["total_memory_bytes"])
    assert sorted(moe.state_dict().keys()) == list(g["state_dict_keys"])
    m = g["m_sizes"]
    offs = np.concatenate([[0], np.cumsum(m)[:-1]])
    outs = moe([t(g["x32"][o:o + c]) for o, c in zip(offs, m)])
    assert outs[3].shape == (0, 256) and outs[3].dtype == torch.float16
    assert np.allclose(torch.cat([o for o in outs if o.shape[0]]).numpy(), g["out32"], atol=1e-5)


def test_moeint4_requires_gpu_kernel():
    """Reference: RuntimeError('CUDA kernel not available') when the extension is missing
    (python/moe_int4_module.py:135-136); here a CPU tensor can never reach the kernel."""
    mod = fq.MoEINT4(2, 64, 32)
    with pytest.raises(RuntimeError):
        mod(torch.zeros(4, 64), torch.zeros(4, dtype=torch.int32), torch.tensor([2, 2], dtype=torch.int32),
            torch.tensor([0, 2], dtype=torch.int32))


def test_gpu_tensors_never_fall_back(monkeypatch):
    """A CUDA tensor with the extension missing must raise, not silently compute on the CPU."""
    from fused_int4_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "SO_PATH", "/nonexistent/libfql_int4.so")
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        _native.lib()


# ---- routing (benchmark/moe_grouped_gemm/routing.py)
def test_routing_matches_reference():
    g = load_golden("f9_routing")
    r = fq.simulate_routing(16, 4, 2, "skewed", device="cpu", seed=42)
    assert np.array_equal(r.expert_indices.numpy(), g["expert_indices"])
    assert np.allclose(r.expert_weights.numpy(), g["expert_weights"], atol=1e-7)
    assert r.tokens_per_expert == list(g["tokens_per_expert"])
    assert r.expert_token_offsets == list(g["expert_token_offsets"])
    rr = fq.simulate_routing(32, 8, 2, "random", device="cpu", seed=7)
    assert np.array_equal(rr.expert_indices.numpy(), g["random_indices"])
    assert rr.tokens_per_expert == list(g["random_tokens_per_expert"])
    x = t(g["x"])
    ein, perm = fq.create_expert_inputs(x, r, 4, 2)
    assert [e.shape[0] for e in ein] == r.tokens_per_expert
    eout = [e * (i + 1.0) for i, e in enumerate(ein)]
    comb = fq.combine_expert_outputs(eout, r, perm, 2)
    assert np.allclose(comb.numpy(), g["combined"], atol=1e-6)
    grouped, tpe, offs, inv = fq.dispatch_grouped(x, r.expert_indices, 4)
    assert tpe.tolist() == r.tokens_per_expert and offs.tolist() == r.expert_token_offsets
    assert torch.equal(grouped, torch.cat(ein))
    with pytest.raises(ValueError):
        fq.simulate_routing(4, 2, 1, "nope", device="cpu")


def test_balanced_routing():
    r = fq.balanced_routing(512, 8, 2, device="cpu", seed=42)
    assert r.tokens_per_expert == [128] * 8
    assert (r.expert_indices[:, 0] != r.expert_indices[:, 1]).all()
    assert torch.allclose(r.expert_weights.sum(1), torch.ones(512))


# ---- additions beyond the reference's surface: host-side behaviour that needs no GPU
def test_new_ops_refuse_cpu_tensors():
    """The product path has no CPU fallback: GPU-only ops raise on host tensors instead of computing something."""
    from fused_int4_amd import ops
    idx = torch.zeros(4, 2, dtype=torch.int64)
    with pytest.raises(RuntimeError):
        ops.route_plan(idx, 4)
    with pytest.raises(RuntimeError):
        ops.combine(torch.zeros(8, 16), torch.zeros(8, dtype=torch.int32), torch.ones(4, 2))
    with pytest.raises(RuntimeError):
        ops.moe_gated_forward(torch.zeros(1, 8, 16, dtype=torch.uint8), torch.ones(1, 8), torch.zeros(1, 8),
                              torch.zeros(3, 64), torch.tensor([3]), torch.tensor([0]))
    with pytest.raises(RuntimeError):
        fq.QuantizedMoEFFN(2, 32, 64)(torch.zeros(4, 32), torch.tensor([2, 2]), torch.tensor([0, 2]))
    with pytest.raises(ValueError):
        ops._precision("bf16")


def test_routing_helpers_cpu_formulation():
    """dispatch_grouped / dispatch_indices / combine_grouped on CPU tensors (the torch formulation the device
    kernels are tested against): round trip through an identity expert returns the routing-weighted tokens."""
    torch.manual_seed(4)
    T, E, top_k, K = 19, 5, 2, 8
    x = torch.randn(T, K)
    route = fq.simulate_routing(T, E, top_k, "random", "cpu", 3)
    grouped, tpe, offs, inv = fq.dispatch_grouped(x, route.expert_indices, E)
    ridx, tpe2, offs2, inv2 = fq.dispatch_indices(route.expert_indices, E)
    assert torch.equal(tpe, tpe2) and torch.equal(offs, offs2) and torch.equal(inv, inv2)
    assert torch.equal(grouped, x[ridx.long()])
    assert int(tpe.sum()) == T * top_k and torch.equal(offs.long(), torch.cumsum(tpe.long(), 0) - tpe.long())
    out = fq.combine_grouped(grouped, route.expert_weights, inv, top_k)
    assert torch.allclose(out, x * route.expert_weights.sum(1, keepdim=True), atol=1e-6)


def test_gated_ffn_oracle_matches_torch_float64():
    """oracle.gated_ffn_grouped (test infrastructure for SURVEY 8f N4) against a direct torch float64 evaluation."""
    torch.manual_seed(8)
    E, H, F = 2, 32, 64
    gate = [torch.randn(F, H) * 0.1 for _ in range(E)]
    up = [torch.randn(F, H) * 0.1 for _ in range(E)]
    down = [torch.randn(H, F) * 0.1 for _ in range(E)]
    ffn = fq.QuantizedMoEFFN.from_weights(gate, up, down)
    assert ffn.gate_up_packed.shape == (E, 2 * F, H // 2) and ffn.down_packed.shape == (E, H, F // 2)
    counts = np.array([3, 5], dtype=np.int32)
    offs = np.array([0, 3], dtype=np.int32)
    x = torch.randn(8, H)
    ref = O.gated_ffn_grouped(
        tuple(b.numpy() for b in (ffn.gate_up_packed, ffn.gate_up_scales, ffn.gate_up_zero_points)),
        tuple(b.numpy() for b in (ffn.down_packed, ffn.down_scales, ffn.down_zero_points)), x.numpy(), counts, offs)
    want = torch.zeros(8, H, dtype=torch.float64)
    for e in range(E):
        wgu = dequantize_weights(ffn.gate_up_packed[e], ffn.gate_up_scales[e], ffn.gate_up_zero_points[e]).double()
        wd = dequantize_weights(ffn.down_packed[e], ffn.down_scales[e], ffn.down_zero_points[e]).double()
        rows = slice(int(offs[e]), int(offs[e] + counts[e]))
        gu = x[rows].double() @ wgu.T
        want[rows] = (torch.nn.functional.silu(gu[:, :F]) * gu[:, F:]) @ wd.T
    assert np.allclose(ref, want.numpy(), rtol=1e-12, atol=1e-12)
    assert ffn.total_memory_bytes == sum(b.numel() * b.element_size() for b in ffn.buffers())


def test_per_group_quantisation_cpu():
    """SURVEY 8f N3 (not in the reference): per-group scales along K = the reference's per-row rule on every group;
    group_size == K reproduces the per-row result bit for bit; the module keeps [N, K / group_size] buffers."""
    import numpy as np
    from oracle import oracle as O
    import fused_int4_amd as fq
    torch.manual_seed(3)
    w = torch.randn(24, 256)
    p, s, z = fq.quantize_weights(w, group_size=64)
    assert p.shape == (24, 128) and s.shape == (24, 4) and z.shape == (24, 4)
    po, so, zo = O.quantize_weights_grouped(w.numpy(), 64)
    assert np.array_equal(p.numpy(), po) and np.array_equal(s.numpy(), so) and np.array_equal(z.numpy(), zo)
    assert np.array_equal(fq.dequantize_weights(p, s, z).numpy(), O.dequantize_weights_grouped(po, so, zo))
    p1, s1, z1 = fq.quantize_weights(w)
    pk, sk, zk = fq.quantize_weights(w, group_size=256)
    assert torch.equal(p1, pk) and torch.equal(s1, sk) and torch.equal(z1, zk)
    # smaller groups -> smaller reconstruction error
    err_row = (fq.dequantize_weights(p1, s1, z1) - w).norm()
    err_grp = (fq.dequantize_weights(p, s, z) - w).norm()
    assert err_grp < err_row
    lin = torch.nn.Linear(256, 24, bias=False)
    m = fq.QuantizedLinear.from_linear(lin, group_size=64)
    assert m.scales.shape == (24, 4) and m.extra_repr().startswith("in_features=256")
    x = torch.randn(5, 256)
    ref = O.reference_linear_grouped(x.numpy(), m.packed_weights.numpy(), m.scales.numpy(), m.zero_points.numpy())
    assert np.allclose(m(x).numpy(), ref, atol=1e-4)
