"""Pin the CPU oracle (oracle/oracle.py and oracle/int4_oracle.c) against golden vectors that
were produced by running the reference's own Python (tests/golden/make_golden.py).

Integer / byte results must be bit-exact; float32 results bit-exact where the arithmetic is
element-wise, 1e-5 where a matmul's summation order is involved (reference tolerance:
tests/test_correctness.py:128)."""
import numpy as np
import pytest

from oracle import oracle as O
from oracle import c_oracle as C
from conftest import load_golden


def test_f1_quantize_dequantize_bit_exact():
    g = load_golden("f1_quant_16x32")
    p, s, z = O.quantize_weights(g["weight"])
    assert np.array_equal(p, g["packed"])
    assert np.array_equal(s, g["scales"])
    assert np.array_equal(z, g["zero_points"])
    assert np.array_equal(O.dequantize_weights(p, s, z), g["dequant"])
    cp, cs, cz = C.quantize_rows(g["weight"])
    assert np.array_equal(cp, g["packed"]) and np.array_equal(cs, g["scales"]) and np.array_equal(cz, g["zero_points"])
    assert np.array_equal(C.dequantize(cp, cs, cz), g["dequant"])
    # reference test tolerance (tests/test_correctness.py:56): round trip within 0.5
    assert np.allclose(g["weight"], g["dequant"], atol=0.5)


@pytest.mark.parametrize("name", ["f2_linear_64x128", "f3_linear_256x512_b4"])
def test_linear_outputs(name):
    g = load_golden(name)
    if "weight" in g:
        p, s, z = O.quantize_weights(g["weight"])
        assert np.array_equal(p, g["packed"]) and np.array_equal(s, g["scales"]) and np.array_equal(z, g["zero_points"])
    out = O.reference_quantized_linear(g["x"], g["packed"], g["scales"], g["zero_points"])
    assert out.shape == g["out"].shape
    assert np.allclose(out, g["out"], atol=1e-5, rtol=1e-5)
    exact = O.reference_quantized_linear(g["x"], g["packed"], g["scales"], g["zero_points"], exact=True)
    assert np.allclose(exact, g["out"], atol=1e-4, rtol=1e-5)
    c64 = C.linear_f64acc(g["x"], g["packed"], g["scales"], g["zero_points"])
    assert np.allclose(c64, exact, atol=1e-9, rtol=1e-12)
    cfma = C.linear_fma(g["x"], g["packed"], g["scales"], g["zero_points"])
    assert np.allclose(cfma, g["out"], atol=1e-3)       # tests/test_correctness.py:218,233


def test_f4_constant_rows():
    g = load_golden("f4_constant_rows")
    for sfx in ("", "2"):
        w = g["weight" + sfx]
        p, s, z = O.quantize_weights(w)
        assert np.array_equal(p, g["packed" + sfx])
        assert np.array_equal(s, g["scales" + sfx])
        assert np.array_equal(z, g["zero_points" + sfx])
        d = O.dequantize_weights(p, s, z)
        assert np.array_equal(d, g["dequant" + sfx])
        assert not np.isnan(d).any()
        cp, cs, cz = C.quantize_rows(w)
        assert np.array_equal(cp, p) and np.array_equal(cs, s) and np.array_equal(cz, z)
    assert np.allclose(g["weight"], g["dequant"], atol=0.5)   # tests/test_correctness.py:103


def test_f5_llm_dims_rows():
    g = load_golden("f5_linear_4096_rows64")
    out = O.reference_quantized_linear(g["x"], g["packed"], g["scales"], g["zero_points"])
    assert np.allclose(out, g["out"], atol=1e-3, rtol=1e-5)
    c64 = C.linear_f64acc(g["x"], g["packed"], g["scales"], g["zero_points"])
    assert np.allclose(c64, g["out"], atol=1e-2)              # tests/test_correctness.py:252
    assert np.abs(c64 - g["out"]).max() < 5e-4


def test_f6_module_vectors():
    g = load_golden("f6_module_128x64")
    p, s, z = O.quantize_weights(g["weight"])
    assert np.array_equal(p, g["packed_weights"]) and np.array_equal(s, g["scales"]) and np.array_equal(z, g["zero_points"])
    assert np.allclose(O.reference_quantized_linear(g["x1"], p, s, z), g["out1"], atol=1e-5)
    assert np.allclose(O.reference_quantized_linear(g["x4"], p, s, z), g["out4"], atol=1e-5)
    assert list(g["state_dict_keys"]) == ["packed_weights", "scales", "zero_points"]


def test_f7_per_tensor_moe_quantiser_bit_exact():
    g = load_golden("f7_moe_per_tensor")
    for sfx in ("", "2"):
        ws = [w for w in g["weights" + sfx]]
        p, s, z = O.quantize_weights_moe(ws)
        assert np.array_equal(p, g["packed" + sfx])
        assert np.array_equal(s, g["scales" + sfx])
        assert np.array_equal(z, g["zero_points" + sfx])
        for e, w in enumerate(ws):
            cp, cs, cz = C.quantize_tensor(w.astype(np.float32))
            assert np.array_equal(cp, p[e]) and cs == s[e, 0] and cz == z[e, 0]


def test_f8_quantized_moe():
    g = load_golden("f8_quantized_moe")
    E = g["weights"].shape[0]
    m = g["m_sizes"]
    packed, scales, zps = [], [], []
    for e in range(E):
        p, s, z = O.quantize_weights(g["weights"][e].astype(np.float32))
        assert np.array_equal(p, g[f"packed{e}"]) and np.array_equal(s, g[f"scales{e}"]) and np.array_equal(z, g[f"zero_points{e}"])
        packed.append(p); scales.append(s); zps.append(z)
    offs = np.concatenate([[0], np.cumsum(m)[:-1]])
    xin = [g["x32"][o:o + c] for o, c in zip(offs, m)]
    outs = O.quantized_moe_forward(xin, packed, scales, zps)
    assert outs[3].shape == (0, 256) and outs[3].dtype == np.float16 and str(g["empty_out_dtype"]) == "torch.float16"
    got = np.concatenate([o for o in outs if o.shape[0]])
    assert np.allclose(got, g["out32"], atol=1e-5)
    # grouped (MoEINT4-style) formulation gives the same rows
    grouped = O.reference_moe_grouped(g["x32"], np.stack(packed), np.stack(scales), np.stack(zps), m, offs)
    assert np.allclose(grouped, g["out32"], atol=1e-5)
    cg = C.moe_grouped(np.stack(packed), np.stack(scales), np.stack(zps), g["x32"], m, offs)
    assert np.allclose(cg, g["out32"], atol=1e-5)
    # fp16 inputs: reference multiplies in fp16; one rounding of the fp32 product is within fp16 eps
    outs16 = O.quantized_moe_forward([x.astype(np.float16) for x in xin], packed, scales, zps)
    got16 = np.concatenate([o for o in outs16 if o.shape[0]]).astype(np.float32)
    assert np.allclose(got16, g["out16"].astype(np.float32), atol=2e-3, rtol=2e-3)
    assert int(g["total_memory_bytes"]) == sum(p.size + 4 * s.size + 4 * z.size for p, s, z in zip(packed, scales, zps))


def test_f9_routing_round_trip():
    g = load_golden("f9_routing")
    grouped, counts, offsets, inv = O.create_expert_inputs(g["x"], g["expert_indices"], 4)
    assert np.array_equal(counts, g["tokens_per_expert"])
    assert np.array_equal(offsets, g["expert_token_offsets"])
    # per-expert row *sets* equal the reference's (its argsort is unstable, order inside an expert is free)
    o = 0
    for c in counts:
        a = grouped[o:o + c]; b = g["expert_inputs"][o:o + c]
        assert np.array_equal(a[np.lexsort(a.T)], b[np.lexsort(b.T)])
        o += c
    eout = np.concatenate([grouped[o:o + c] * (e + 1.0) for e, (o, c) in enumerate(zip(offsets, counts))])
    comb = O.combine_expert_outputs(eout, g["expert_weights"], inv, 2)
    assert np.allclose(comb, g["combined"], atol=1e-6)


def test_reference_byte_model():
    # benchmark/run_benchmark.py:222,227 and docs/runpod-guide.md:220 (22.20 MB at 4096x11008)
    b, f = O.reference_roofline_model(4096, 11008)
    assert b == 22_648_832 and f == 90_177_536


def test_c_unpack_matches_python():
    rng = np.random.default_rng(0)
    p = rng.integers(0, 256, size=(7, 33), dtype=np.uint8)
    assert np.array_equal(C.unpack(p), O.unpack_nibbles(p))


def test_e4m3_format_is_pinned_to_torch():
    """fp8 activations are not in the reference (README.md:228), so the oracle's e4m3 table and rounding are pinned to
    torch's float8_e4m3fn casts instead: all 256 codes, and round-to-nearest-even on values that include every exact
    midpoint, the subnormals, the saturation edge (448 / 464) and signed zeros."""
    import torch
    tab = O.e4m3_table()
    tt = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    assert np.array_equal(np.isnan(tab), np.isnan(tt))
    assert np.array_equal(tab[~np.isnan(tab)], tt[~np.isnan(tt)])
    rng = np.random.default_rng(0)
    fin = tab[:127]
    x = np.concatenate([rng.standard_normal(20000).astype(np.float32) * s for s in (1e-3, 0.01, 1, 30, 200)]
                       + [fin, -fin, (fin[:-1] + fin[1:]) / 2, np.array([448, 460, 463.9, 464, -464, 1e-9, -0.0, 0.0], np.float32)]).astype(np.float32)
    enc = O.e4m3_encode(x)
    tenc = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    assert np.array_equal(enc, tenc)
    assert np.array_equal(O.e4m3_decode(enc[:1000]), torch.from_numpy(x[:1000]).to(torch.float8_e4m3fn).float().numpy())
    # the library's per-row quantisation rule == the torch-side helper a caller would use
    xr = rng.standard_normal((7, 96)).astype(np.float32)
    b, sc = O.quantize_activations_fp8(xr)
    t = torch.from_numpy(xr)
    amax = t.abs().amax(dim=1)
    assert np.array_equal(sc, (amax / 448.0).numpy())
    assert np.array_equal(b, (t / (amax / 448.0)[:, None]).to(torch.float8_e4m3fn).view(torch.uint8).numpy())


def test_torch_op_oracle_is_pinned_too():
    """oracle/oracle_torch.py (the form bench.py's cpu_baseline times: the reference's own tensor ops) against the golden
    vectors and against the numpy restatement."""
    import torch
    from oracle import oracle_torch as OT
    g = load_golden("f1_quant_16x32")
    d = OT.dequantize_weights(torch.from_numpy(g["packed"]), torch.from_numpy(g["scales"]), torch.from_numpy(g["zero_points"]))
    assert np.array_equal(d.numpy(), g["dequant"])
    for name in ("f2_linear_64x128", "f3_linear_256x512_b4"):
        g = load_golden(name)
        args = [torch.from_numpy(np.ascontiguousarray(g[k])) for k in ("x", "packed", "scales", "zero_points")]
        out = OT.reference_quantized_linear(*args).numpy()
        assert out.shape == g["out"].shape
        assert np.allclose(out, g["out"], atol=1e-5, rtol=1e-5)
        assert np.allclose(out, O.reference_quantized_linear(g["x"], g["packed"], g["scales"], g["zero_points"]), atol=1e-5, rtol=1e-5)
