"""GPU parity tests (run with `-m gpu` on an MI355X).  Every call goes through the C ABI of
csrc/libfql_int4.so (via the ctypes shim in ops.py); the checker is the CPU oracle (oracle/) and
the golden vectors generated from the reference's own Python (tests/golden/).

Bars: bit-exact for the INT4 unpack / index / limb paths and for integer-valued inputs;
floating-point outputs within the tolerances written in tests/helpers.py, and within the
reference's own test tolerances (tests/test_correctness.py:218,233,252: atol 1e-3 / 1e-3 / 1e-2)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from helpers import (EXACT_REL_FRO, FAST_REL_FRO, INT8_REL_FRO, INT8_REL_FRO_LARGE_K, FMA_REL_FRO, rel_fro, act_limbs_reference,
                     decode_limbs)
from oracle import oracle as O
from oracle import c_oracle as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fq():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import fused_int4_amd as pkg
    from fused_int4_amd import _native
    _native.lib()                       # fails loudly if the extension is not built
    return pkg


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def make_problem(N, K, B, seed, wscale=1.0):
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((N, K)) * wscale).astype(np.float32)
    x = rng.standard_normal((B, K)).astype(np.float32)
    p, s, z = O.quantize_weights(w)
    return x, p, s, z


# ------------------------------------------------------------------------------ format helpers
def test_unpack_bit_exact(fq):
    from fused_int4_amd import ops
    rng = np.random.default_rng(0)
    for shape in [(1, 1), (3, 5), (64, 2048), (7, 33), (11008, 64)]:
        p = rng.integers(0, 256, size=shape, dtype=np.uint8)
        got = ops.unpack_nibbles(dev(p)).cpu().numpy()
        assert np.array_equal(got, O.unpack_nibbles(p)), shape
        assert np.array_equal(got, C.unpack(p)), shape


def test_dequantize_bit_exact(fq):
    from fused_int4_amd import ops
    g = load_golden("f1_quant_16x32")
    got = ops.dequantize_forward(dev(g["packed"]), dev(g["scales"]), dev(g["zero_points"])).cpu().numpy()
    assert np.array_equal(got, g["dequant"])
    x, p, s, z = make_problem(300, 1030, 1, 3)
    got = ops.dequantize_forward(dev(p), dev(s), dev(z)).cpu().numpy()
    assert np.array_equal(got, O.dequantize_weights(p, s, z))
    # the package-level API dispatches to the same kernel on GPU tensors
    got2 = fq.dequantize_weights(dev(p), dev(s), dev(z)).cpu().numpy()
    assert np.array_equal(got2, got)


@pytest.mark.parametrize("L,prec", [(3, "exact"), (2, "fast"), (1, "int8")])
def test_activation_limbs_bit_exact(fq, L, prec):
    """The pre-pass (phase 1) against a numpy restatement of the same fixed-point rule."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(5)
    T, K = 37, 544                                          # K % 256 != 0 -> zero padded; ragged T
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[3] = 0.0                                              # all-zero row
    x[4] *= 1e-30                                           # tiny magnitudes
    x[5] *= 1e30                                            # huge magnitudes
    x[6, 7] = 2.0 ** 10                                     # row max exactly a power of two
    x[7, :] = np.float32(32639.6)                           # rounds above the 2-limb limit -> exponent bump
    limbs, delta, rowsum = ops.act_quant(dev(x), precision=prec)
    Kp = 768
    dig, covered = decode_limbs(limbs.cpu().numpy(), L, T, 1, K, Kp)
    ref_dig, ref_delta, ref_sum = act_limbs_reference(x, L)
    assert covered.all()
    assert np.array_equal(delta[0].cpu().numpy(), ref_delta)
    assert np.array_equal(dig[:, :, :K], ref_dig)
    assert (dig[:, :, K:] == 0).all()
    assert np.array_equal(rowsum[0].cpu().numpy(), ref_sum)
    assert dig.min() >= -128 and dig.max() <= 127
    # reconstruction error: at most half a unit of the last limb
    X = sum(ref_dig[l] * 256 ** l for l in range(L))
    assert np.all(np.abs(x.astype(np.float64) - X * ref_delta[:, None].astype(np.float64)) <= 0.5 * ref_delta[:, None] * (1 + 1e-7))


def test_activation_limbs_grouped_layout(fq):
    from fused_int4_amd import ops
    rng = np.random.default_rng(6)
    T, K = 50, 256
    x = rng.standard_normal((T, K)).astype(np.float32)
    counts = np.array([7, 0, 33, 5], dtype=np.int32)        # 5 trailing rows uncovered
    offs = np.array([0, 7, 7, 40], dtype=np.int32)
    limbs, delta, rowsum = ops.act_quant(dev(x), precision="exact", tokens_per_expert=dev(counts), input_offsets=dev(offs))
    dig, covered = decode_limbs(limbs.cpu().numpy(), 3, T, 4, K, 256, counts, offs)
    ref_dig, ref_delta, _ = act_limbs_reference(x, 3)
    assert covered.sum() == 45
    assert np.array_equal(dig[:, covered], ref_dig[:, covered])
    assert np.array_equal(delta[0].cpu().numpy()[covered], ref_delta[covered])


def test_gpu_quantisers_bit_exact(fq):
    """weights quantised on the GPU == weights quantised by the reference on the host."""
    from fused_int4_amd import ops
    for name in ("f1_quant_16x32", "f2_linear_64x128"):
        g = load_golden(name)
        p, s, z = ops.quantize_rows(dev(g["weight"]))
        assert np.array_equal(p.cpu().numpy(), g["packed"])
        assert np.array_equal(s.cpu().numpy(), g["scales"]) and np.array_equal(z.cpu().numpy(), g["zero_points"])
    g = load_golden("f4_constant_rows")
    for sfx in ("", "2"):
        p, s, z = fq.quantize_weights(dev(g["weight" + sfx]))           # package API dispatches to the HIP kernel
        assert np.array_equal(p.cpu().numpy(), g["packed" + sfx])
        assert np.array_equal(s.cpu().numpy(), g["scales" + sfx]) and np.array_equal(z.cpu().numpy(), g["zero_points" + sfx])
    rng = np.random.default_rng(8)
    w = (rng.standard_normal((300, 1030)) * 0.02).astype(np.float32)
    w[5] = 0.5; w[6, ::2] = -1.0; w[6, 1::2] = 1.0
    p, s, z = ops.quantize_rows(dev(w))
    rp, rs, rz = O.quantize_weights(w)
    assert np.array_equal(p.cpu().numpy(), rp) and np.array_equal(s.cpu().numpy(), rs) and np.array_equal(z.cpu().numpy(), rz)
    g = load_golden("f7_moe_per_tensor")
    for sfx in ("", "2"):
        pk, sc, zp = fq.quantize_weights_moe([torch.from_numpy(x).cuda() for x in g["weights" + sfx]])
        assert np.array_equal(pk.cpu().numpy(), g["packed" + sfx])
        assert np.array_equal(sc.cpu().numpy(), g["scales" + sfx]) and np.array_equal(zp.cpu().numpy(), g["zero_points" + sfx])


# ------------------------------------------------------------------------------ linear: reference's own cases
def test_golden_f2_64x128_1d(fq):
    """tests/test_correctness.py:201-219 (atol=1e-3) through the drop-in operator name."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(fq.__file__), "dropin"))
    import fused_quant_linear_cuda as ext
    g = load_golden("f2_linear_64x128")
    out = ext.forward(dev(g["x"]), dev(g["packed"]), dev(g["scales"]), dev(g["zero_points"]))
    assert out.shape == (64,)
    assert torch.allclose(torch.from_numpy(g["out"]), out.cpu(), atol=1e-3)
    assert rel_fro(out.cpu().numpy(), g["out"]) < FMA_REL_FRO


def test_golden_f3_256x512_b4(fq):
    """tests/test_correctness.py:221-234 (atol=1e-3)."""
    from fused_int4_amd import ops
    g = load_golden("f3_linear_256x512_b4")
    out = ops.linear_forward(dev(g["x"]), dev(g["packed"]), dev(g["scales"]), dev(g["zero_points"]))
    assert out.shape == (4, 256)
    assert torch.allclose(torch.from_numpy(g["out"]), out.cpu(), atol=1e-3)


def test_golden_f5_4096x4096_rows(fq):
    """tests/test_correctness.py:236-253 (atol=1e-2), the 64-row slice kept in the fixture."""
    from fused_int4_amd import ops
    g = load_golden("f5_linear_4096_rows64")
    out = ops.linear_forward(dev(g["x"]), dev(g["packed"]), dev(g["scales"]), dev(g["zero_points"]))
    assert torch.allclose(torch.from_numpy(g["out"]), out.cpu(), atol=1e-2)
    assert np.abs(out.cpu().numpy() - g["out"]).max() < 1e-3


def test_golden_f6_module(fq):
    g = load_golden("f6_module_128x64")
    lin = torch.nn.Linear(128, 64, bias=False)
    lin.weight.data = torch.from_numpy(g["weight"])
    ql = fq.QuantizedLinear.from_linear(lin).cuda()
    assert sorted(ql.state_dict().keys()) == list(g["state_dict_keys"])
    assert np.array_equal(ql.packed_weights.cpu().numpy(), g["packed_weights"])
    o1 = ql(dev(g["x1"]))
    o4 = ql(dev(g["x4"]))
    assert o1.shape == (64,) and o4.shape == (4, 64)
    assert torch.allclose(torch.from_numpy(g["out1"]), o1.cpu(), atol=1e-3)
    assert torch.allclose(torch.from_numpy(g["out4"]), o4.cpu(), atol=1e-3)
    assert ql.extra_repr() == str(g["extra_repr"])


# ------------------------------------------------------------------------------ linear: every kernel path
@pytest.mark.parametrize("B", [1, 2, 3, 4])
@pytest.mark.parametrize("N,K", [(64, 128), (1000, 512), (11008, 4096)])
def test_gemv_path(fq, B, N, K):
    from fused_int4_amd import ops
    x, p, s, z = make_problem(N, K, B, 100 + B)
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    assert rel_fro(out, ref) < FMA_REL_FRO
    assert np.allclose(out, O.reference_quantized_linear(x, p, s, z), atol=1e-2 if K > 1024 else 1e-3, rtol=1e-5)


@pytest.mark.parametrize("B,N,K", [(5, 64, 128), (8, 96, 32), (33, 200, 96), (64, 192, 256), (100, 1000, 544),
                                   (128, 256, 1024), (129, 193, 2048), (300, 400, 320), (512, 384, 4096)])
@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO)])
def test_mfma_path_shapes(fq, B, N, K, prec, tol):
    """B > 4 with K % 32 == 0: activation pre-pass + MFMA GEMM; ragged M, N and K tails."""
    from fused_int4_amd import ops
    x, p, s, z = make_problem(N, K, B, B + N)
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec).cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    assert out.shape == ref.shape
    assert rel_fro(out, ref) < tol
    if prec == "exact":      # the reference's GPU test tolerance, at every shape
        assert np.allclose(out, O.reference_quantized_linear(x, p, s, z), atol=1e-3 if K <= 1024 else 1e-2, rtol=1e-5)


@pytest.mark.parametrize("B,N,K", [(1, 10, 2), (3, 17, 6), (8, 5, 34), (20, 33, 66), (6, 64, 4098)])
def test_generic_path_odd_shapes(fq, B, N, K):
    """K % 32 != 0 (the reference only requires even K): shape-generic fused kernel."""
    from fused_int4_amd import ops
    x, p, s, z = make_problem(N, K, B, 7 * B + K)
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    assert rel_fro(out, C.linear_f64acc(x, p, s, z)) < FMA_REL_FRO


def test_integer_inputs_are_bit_exact(fq):
    """Integer-valued activations make the fixed-point conversion exact, so the MFMA path must reproduce
    float32(scale * float32(integer dot product)) bit for bit -- the INT4 unpack, the k permutation,
    the fragment layouts and the limb recombination all have to be right for that."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(11)
    B, N, K = 96, 320, 1024
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    p = ((q[:, 1::2] << 4) | q[:, 0::2]).astype(np.uint8)
    zp = rng.integers(0, 16, size=N).astype(np.float32)
    sc = (rng.random(N).astype(np.float32) + 0.5) * np.float32(0.01)
    x = rng.integers(-2000, 2001, size=(B, K)).astype(np.float32)
    x[:, 0] = 2048.0                                        # row max = 2^11 -> delta = 2^-11 exactly divides
    exact_int = x.astype(np.int64) @ (q.astype(np.int64) - zp.astype(np.int64)[:, None]).T      # [B,N]
    assert np.abs(exact_int).max() < 2 ** 53
    # the kernel's own last steps: float32 Horner of exact limb sums, * delta, * scale
    out = ops.linear_forward(dev(x), dev(p), dev(sc), dev(zp), precision="exact").cpu().numpy()
    ref = (exact_int.astype(np.float64) * sc.astype(np.float64)[None, :])
    assert rel_fro(out, ref) < 3e-7
    # values small enough to be exactly representable end to end -> bit exact
    x2 = rng.integers(-7, 8, size=(B, K)).astype(np.float32)
    ei = x2.astype(np.int64) @ (q.astype(np.int64) - zp.astype(np.int64)[:, None]).T
    assert np.abs(ei).max() < 2 ** 24
    out2 = ops.linear_forward(dev(x2), dev(p), dev(sc), dev(zp), precision="exact").cpu().numpy()
    assert np.array_equal(out2, (ei.astype(np.float32) * sc[None, :]).astype(np.float32))
    out3 = ops.linear_forward(dev(x2), dev(p), dev(sc), dev(zp), precision="fast").cpu().numpy()
    assert np.array_equal(out3, out2)


def test_non_integer_zero_points(fq):
    """The format stores zero_points as float32; nothing forces them to be integers."""
    from fused_int4_amd import ops
    x, p, s, z = make_problem(128, 256, 40, 21)
    z = (z + 0.37).astype(np.float32)
    for B in (2, 40):
        out = ops.linear_forward(dev(x[:B]), dev(p), dev(s), dev(z)).cpu().numpy()
        assert rel_fro(out, C.linear_f64acc(x[:B], p, s, z)) < 5e-6


def test_special_rows(fq):
    from fused_int4_amd import ops
    x, p, s, z = make_problem(64, 256, 16, 22)
    x[0] = 0.0
    x[1] *= 1e-20
    x[2] *= 1e20
    x[3, 5] = np.inf
    x[4, 9] = np.nan
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    ref = C.linear_f64acc(x[:3], p, s, z)
    assert (out[0] == 0).all()
    assert rel_fro(out[1], ref[1]) < EXACT_REL_FRO and rel_fro(out[2], ref[2]) < EXACT_REL_FRO
    assert np.isnan(out[3]).all() and np.isnan(out[4]).all()      # documented: a non-finite activation poisons its row
    assert np.isfinite(out[5:]).all()


@pytest.mark.parametrize("B,N,K", [(1, 200, 256), (3, 64, 128), (4, 96, 128), (40, 200, 544), (300, 400, 320), (6, 33, 66), (20, 192, 4096)])
def test_bias_in_every_linear_path(fq, B, N, K):
    """SURVEY 8f N4 (the reference asserts `bias is None`, python/module.py:84): GEMV, MFMA (wide / 32-row / 16-row
    tiles by batch size) and the generic kernel all add the bias as their last operation."""
    from fused_int4_amd import ops
    x, p, s, z = make_problem(N, K, B, 31 * B + N)
    bias = np.random.default_rng(B).standard_normal(N).astype(np.float32)
    base = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    got = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), bias=dev(bias)).cpu().numpy()
    assert np.array_equal(got, base + bias[None, :])            # one float32 add after the un-biased result
    assert rel_fro(got, C.linear_f64acc(x, p, s, z) + bias[None, :].astype(np.float64)) < 5e-6
    lin = torch.nn.Linear(K, N, bias=True)
    m = fq.QuantizedLinear.from_linear(lin).cuda()
    assert sorted(m.state_dict().keys()) == ["bias", "packed_weights", "scales", "zero_points"]
    xt = torch.from_numpy(x)
    want = m.cpu()(xt)                                          # CPU path: dequantize-then-matmul + bias
    got_m = m.cuda()(xt.cuda()).cpu()
    assert torch.allclose(got_m, want, atol=1e-3 if K <= 1024 else 1e-2, rtol=1e-5)


# ------------------------------------------------------------------------------ error behaviour (csrc/quantized_linear_kernel.cu:311-335)
def test_linear_errors(fq):
    from fused_int4_amd import ops
    x, p, s, z = make_problem(64, 128, 2, 1)
    X, P, S, Z = dev(x), dev(p), dev(s), dev(z)
    with pytest.raises(RuntimeError, match="input must be a CUDA tensor"):
        ops.linear_forward(torch.from_numpy(x), P, S, Z)
    with pytest.raises(RuntimeError, match="packed_weights must be a CUDA tensor"):
        ops.linear_forward(X, torch.from_numpy(p), S, Z)
    with pytest.raises(RuntimeError, match="input must be contiguous"):
        ops.linear_forward(dev(np.zeros((128, 2), np.float32)).t(), P, S, Z)
    with pytest.raises(RuntimeError, match="input must be float32"):
        ops.linear_forward(X.double(), P, S, Z)
    with pytest.raises(RuntimeError, match="packed_weights must be uint8"):
        ops.linear_forward(X, P.to(torch.int32), S, Z)
    with pytest.raises(RuntimeError, match="scales must be float32"):
        ops.linear_forward(X, P, S.double(), Z)
    with pytest.raises(RuntimeError, match="packed_weights dim 1 must be input_dim / 2"):
        ops.linear_forward(X[:, :64].contiguous(), P, S, Z)
    with pytest.raises(RuntimeError, match="output_dim elements"):
        ops.linear_forward(X, P, S[:10], Z)
    with pytest.raises(ValueError):
        ops.linear_forward(X, P, S, Z, precision="bf16")
    # empty batch
    assert ops.linear_forward(X[:0], P, S, Z).shape == (0, 64)


# ------------------------------------------------------------------------------ MoE
def make_moe(E, N, K, counts, seed, gap_rows=0, wscale=0.02):
    rng = np.random.default_rng(seed)
    P, S, Z = [], [], []
    for _ in range(E):
        p, s, z = O.quantize_weights((rng.standard_normal((N, K)) * wscale).astype(np.float32))
        P.append(p); S.append(s); Z.append(z)
    counts = np.asarray(counts, dtype=np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    T = int(counts.sum()) + gap_rows
    x = rng.standard_normal((T, K)).astype(np.float32)
    return np.stack(P), np.stack(S), np.stack(Z), x, counts, offs


@pytest.mark.parametrize("E,N,K,counts,gap", [
    (4, 256, 128, [40, 0, 33, 20], 3),
    (8, 384, 512, [128] * 8, 0),
    (8, 200, 256, [300, 1, 0, 17, 129, 64, 0, 5], 11),
    (3, 64, 96, [5, 6, 7], 0),                               # K % 256 != 0
    (2, 130, 34, [9, 4], 2),                                 # K % 32 != 0 -> generic grouped kernel
    (16, 192, 256, [0] * 15 + [50], 0),
    (100, 64, 64, [(7 * i) % 5 for i in range(100)], 4),     # > 64 experts: multi-chunk device-side expert table
    (70, 96, 128, [0] * 64 + [33, 0, 1, 40, 0, 2], 0),       # all the work behind the first 64-expert chunk
])
@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO)])
def test_moe_grouped_parity(fq, E, N, K, counts, gap, prec, tol):
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(fq.__file__), "dropin"))
    import moe_int4_cuda as ext
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, sum(counts) + E, gap_rows=gap)
    T = x.shape[0]
    expert_ids = torch.zeros(T, dtype=torch.int32).cuda()    # accepted, ignored (reference :98)
    if prec == "exact":
        out = ext.forward(dev(P), dev(S), dev(Z), dev(x), expert_ids, dev(cnt), dev(offs)).cpu().numpy()
    else:
        from fused_int4_amd import ops
        out = ops.moe_forward(dev(P), dev(S), dev(Z), dev(x), None, dev(cnt), dev(offs), precision=prec).cpu().numpy()
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert out.shape == (T, N)
    assert rel_fro(out, ref) < (tol if K % 32 == 0 else FMA_REL_FRO)
    covered = np.zeros(T, bool)
    for c, o in zip(cnt, offs):
        covered[o:o + c] = True
    assert (out[~covered] == 0).all()                        # torch::zeros semantics (reference :109)


@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO)])
def test_every_tile_configuration_is_bit_identical(fq, prec, tol):
    """The integer dot products are exact, so the tile shape, the K split inside a workgroup and the row
    grouping must not change a single output bit: sweep every compiled configuration (wide tiles 0..,
    short-row-group tiles 100.., decode-size 16-row tiles 200..) through the tuning entry point on a ragged grouped problem."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    E, N, K = 5, 200, 768                                    # N % 32 != 0, three 256-k stages
    counts = [0, 7, 33, 70, 129]
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, 123)
    T = x.shape[0]
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision=prec, tokens_per_expert=dc, input_offsets=do)
    outs = {}
    cfgs = [c for c in (list(range(lib.fql_tune_num_configs())) + list(range(100, 100 + lib.fql_tune_num_rows32_configs()))
                        + list(range(200, 200 + lib.fql_tune_num_rows16_configs()))
                        + list(range(300, 300 + lib.fql_tune_num_w4_configs())))
            if lib.fql_tune_is_config(c, ops._precision(prec))]      # (not every wide id is built for every limb count)
    assert len(cfgs) >= 20 and cfgs[0] in (0, 1)
    for cfg in cfgs:
        out = torch.full((T, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = ops.tune_gemm_i8(cfg, limbs, delta, rowsum, dP, dS, dZ, dc, do, out, E, T, K, N, prec)
        assert rc == 0, (cfg, rc)
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy()
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert rel_fro(outs[cfgs[0]], ref) < tol
    for cfg, o in outs.items():
        assert np.array_equal(o, outs[cfgs[0]]), f"configuration {cfg} differs from configuration {cfgs[0]}"


def test_moe_equals_per_expert_linear_bitwise(fq):
    """Integer-exact accumulation: the grouped launch and E separate linear launches (other tile
    shapes, other row offsets) must agree bit for bit."""
    from fused_int4_amd import ops
    P, S, Z, x, cnt, offs = make_moe(4, 320, 512, [70, 130, 8, 200], 77)
    out = ops.moe_forward(dev(P), dev(S), dev(Z), dev(x), None, dev(cnt), dev(offs)).cpu().numpy()
    for e in range(4):
        o, c = int(offs[e]), int(cnt[e])
        single = ops.linear_forward(dev(x[o:o + c]), dev(P[e]), dev(S[e]), dev(Z[e])).cpu().numpy()
        assert np.array_equal(single, out[o:o + c]), e


def test_moe_golden_f8_quantized_moe(fq):
    """QuantizedMoE (benchmark/moe_grouped_gemm/moe_int4_module.py) vectors from the reference."""
    g = load_golden("f8_quantized_moe")
    ws = [torch.from_numpy(w) for w in g["weights"]]
    moe = fq.QuantizedMoE.from_fp16_weights(ws).cuda()
    for e in range(4):
        assert np.array_equal(moe.experts[e].packed_weights.cpu().numpy(), g[f"packed{e}"])
    assert moe.total_memory_bytes == int(g["total_memory_bytes"])
    assert sorted(moe.state_dict().keys()) == list(g["state_dict_keys"])
    m = g["m_sizes"]
    offs = np.concatenate([[0], np.cumsum(m)[:-1]])
    xin = [dev(g["x32"][o:o + c]) for o, c in zip(offs, m)]
    outs = moe(xin)
    assert outs[3].shape == (0, 256) and outs[3].dtype == torch.float16      # reference quirk :65-68
    got = torch.cat([o for o in outs if o.shape[0]]).cpu().numpy()
    assert np.allclose(got, g["out32"], atol=1e-5)
    outs16 = moe([t.half() for t in xin])
    assert outs16[0].dtype == torch.float16
    got16 = torch.cat([o for o in outs16 if o.shape[0]]).float().cpu().numpy()
    assert np.allclose(got16, g["out16"].astype(np.float32), atol=2e-3, rtol=2e-3)


def test_moeint4_module_per_tensor(fq):
    """MoEINT4 (python/moe_int4_module.py:83-146): per-tensor quantiser + grouped forward."""
    g = load_golden("f7_moe_per_tensor")
    mod = fq.MoEINT4.from_weights([torch.from_numpy(w).cuda() for w in g["weights"]])     # HIP quantiser
    assert np.array_equal(mod.packed_weights.cpu().numpy(), g["packed"])
    assert np.array_equal(mod.scales.cpu().numpy(), g["scales"])
    assert np.array_equal(mod.zero_points.cpu().numpy(), g["zero_points"])
    assert sorted(mod.state_dict().keys()) == list(g["state_dict_keys"])
    rng = np.random.default_rng(3)
    x = rng.standard_normal((24, 64)).astype(np.float32)
    cnt = np.array([10, 14], np.int32); offs = np.array([0, 10], np.int32)
    out = mod(dev(x), torch.zeros(24, dtype=torch.int32).cuda(), dev(cnt), dev(offs)).cpu().numpy()
    ref = C.moe_grouped(g["packed"], g["scales"], g["zero_points"], x, cnt, offs)
    assert rel_fro(out, ref) < EXACT_REL_FRO


def test_fused_dispatch_gather_equals_materialised(fq):
    """gather fused into the pre-pass == index_select + grouped GEMM, bit for bit, and the
    whole dispatch -> GEMM -> combine round trip matches the CPU oracle of routing.py semantics."""
    from fused_int4_amd import ops
    E, K, N, Ttok, top_k = 8, 512, 384, 96, 2
    P, S, Z, _, _, _ = make_moe(E, N, K, [1] * E, 31)
    r = fq.simulate_routing(Ttok, E, top_k, "skewed", device="cuda", seed=3)
    x = torch.randn(Ttok, K, device="cuda")
    grouped, tpe, offs, inv = fq.dispatch_grouped(x, r.expert_indices, E)
    ref = ops.moe_forward(dev(P), dev(S), dev(Z), grouped.contiguous(), None, tpe, offs)
    ri, tpe2, offs2, inv2 = fq.dispatch_indices(r.expert_indices, E)
    got = ops.moe_gather_forward(dev(P), dev(S), dev(Z), x, ri, tpe2, offs2)
    assert torch.equal(tpe, tpe2) and torch.equal(offs, offs2) and torch.equal(inv, inv2)
    assert torch.equal(got, ref)
    y = fq.combine_grouped(got, r.expert_weights, inv2, top_k).cpu().numpy()
    gx, cnt, of, oinv = O.create_expert_inputs(x.cpu().numpy(), r.expert_indices.cpu().numpy(), E)
    oy = C.moe_grouped(P, S, Z, gx, cnt, of)
    yref = O.combine_expert_outputs(oy, r.expert_weights.cpu().numpy(), oinv, top_k)
    assert rel_fro(y, yref) < 5e-6
    # out-of-range indices are clamped, not dereferenced
    bad = ri.clone(); bad[0] = 10 ** 6; bad[1] = -5
    ops.moe_gather_forward(dev(P), dev(S), dev(Z), x, bad, tpe2, offs2)
    torch.cuda.synchronize()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("prec", ["exact", "fast"])
def test_native_16bit_io_equals_converted_float32(fq, dtype, prec):
    """float16 / bfloat16 activations in and out.  The kernels widen the inputs exactly and round
    the float32 result to nearest even, so the native path must equal float32-path(x.float()).to(dtype) bit for
    bit -- linear (wide tiles, ragged N), grouped MoE with an empty expert, uncovered rows and few-row groups."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(11)
    # linear, MFMA path
    x, p, s, z = make_problem(200, 512, 70, 5)
    xd = dev(x).to(dtype)
    got = ops.linear_forward_any(xd, dev(p), dev(s), dev(z), precision=prec)
    want = ops.linear_forward(xd.float(), dev(p), dev(s), dev(z), precision=prec).to(dtype)
    assert got.dtype == dtype and torch.equal(got, want)
    # float32 in, 16-bit out and the reverse
    got = ops.linear_forward_any(dev(x), dev(p), dev(s), dev(z), precision=prec, out_dtype=dtype)
    assert torch.equal(got, ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec).to(dtype))
    got = ops.linear_forward_any(xd, dev(p), dev(s), dev(z), precision=prec, out_dtype=torch.float32)
    assert torch.equal(got, ops.linear_forward(xd.float(), dev(p), dev(s), dev(z), precision=prec))
    # B <= 4 and K % 32 != 0 fall back to conversion, same contract
    x1, p1, s1, z1 = make_problem(64, 96, 3, 6)
    got = ops.linear_forward_any(dev(x1).to(dtype), dev(p1), dev(s1), dev(z1), precision=prec)
    assert torch.equal(got, ops.linear_forward(dev(x1).to(dtype).float(), dev(p1), dev(s1), dev(z1), precision=prec).to(dtype))
    # grouped, with an empty expert, a 3-row gap of uncovered rows; wide tiles and the short-row-group kernel
    for counts in ([40, 0, 133, 20], [3, 7, 0, 30, 1, 12, 9, 2]):
        E = len(counts)
        P, S, Z, xm, cnt, offs = make_moe(E, 264, 256, counts, 17 + E, gap_rows=3)
        xmd = dev(xm).to(dtype)
        got = ops.moe_forward_any(dev(P), dev(S), dev(Z), xmd, None, dev(cnt), dev(offs), precision=prec)
        want = ops.moe_forward(dev(P), dev(S), dev(Z), xmd.float(), None, dev(cnt), dev(offs), precision=prec).to(dtype)
        assert got.dtype == dtype and torch.equal(got, want)
        assert (got[-3:] == 0).all()


def test_quantized_moe_module_keeps_16bit_inputs_native(fq):
    """QuantizedMoE fed float16 rows (as the reference's benches do) returns float16 and matches the CPU oracle."""
    torch.manual_seed(3)
    E, K, N = 4, 128, 96
    ws = [torch.randn(N, K) * 0.02 for _ in range(E)]
    moe = fq.QuantizedMoE.from_fp16_weights([w.half() for w in ws]).cuda()
    xs = [torch.randn(m, K).half() for m in (9, 0, 33, 5)]
    outs = moe([x.cuda() for x in xs])
    for e, (x, o) in enumerate(zip(xs, outs)):
        assert o.dtype == torch.float16
        if x.shape[0] == 0:
            continue
        p, s, z = (t.cpu().numpy() for t in (moe.experts[e].packed_weights, moe.experts[e].scales, moe.experts[e].zero_points))
        ref = C.linear_f64acc(x.float().numpy(), p, s, z)
        assert rel_fro(o.float().cpu().numpy(), ref) < 1e-3       # float16 rounding of the outputs


@pytest.mark.parametrize("T,E,top_k", [(16, 4, 2), (512, 8, 2), (333, 64, 6), (1, 3, 1), (4097, 128, 4)])
def test_route_plan_is_the_stable_sort(fq, T, E, top_k):
    """fql_route_plan_i32 == argsort(stable=True) / bincount / cumsum of routing.py:117-149, index for index."""
    from fused_int4_amd import ops
    g = torch.Generator().manual_seed(T * 31 + E)
    idx = torch.stack([torch.randperm(E, generator=g)[:top_k] for _ in range(T)]) if top_k <= E else None
    counts, offsets, token_of_sorted, pos_of_slot = ops.route_plan(idx.cuda(), E)
    flat = idx.reshape(-1)
    order = torch.argsort(flat, stable=True)
    ref_counts = torch.bincount(flat, minlength=E)
    assert torch.equal(counts.cpu().long(), ref_counts)
    assert torch.equal(offsets.cpu().long(), torch.cumsum(ref_counts, 0) - ref_counts)
    assert torch.equal(token_of_sorted.cpu().long(), order // top_k)
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(order.numel())
    assert torch.equal(pos_of_slot.cpu().long(), inverse)


@pytest.mark.parametrize("top_k", [1, 2, 4])
def test_combine_kernel(fq, top_k):
    from fused_int4_amd import ops
    torch.manual_seed(5)
    T, N = 37, 1004
    y = torch.randn(T * top_k, N)
    pos = torch.randperm(T * top_k).to(torch.int32)
    w = torch.rand(T, top_k) + 0.1
    got = ops.combine(y.cuda(), pos.cuda(), w.cuda()).cpu()
    ref = (y[pos.long()].view(T, top_k, N) * w.unsqueeze(-1)).sum(dim=1)
    if top_k <= 2:
        assert torch.equal(got, ref)                         # one addition: no ordering freedom
    else:
        assert torch.allclose(got, ref, rtol=1e-6, atol=1e-6)
    got = ops.combine(y[:, :1001].contiguous().cuda(), pos.cuda(), w.cuda()).cpu()      # N % 4 != 0
    assert torch.allclose(got, (y[pos.long()][:, :1001].reshape(T, top_k, 1001) * w.unsqueeze(-1)).sum(dim=1), rtol=1e-6, atol=1e-6)


def test_regroup_index(fq):
    from fused_int4_amd import ops
    torch.manual_seed(9)
    G, EL = 4, 3
    cnt = torch.randint(0, 7, (G, EL))
    cnt[1, 2] = 0
    R = int(cnt.sum())
    tpe, offs, gather, scatter = ops.regroup_index(cnt.cuda(), R)
    assert torch.equal(tpe.cpu().long(), cnt.sum(0))
    assert torch.equal(offs.cpu().long(), torch.cumsum(cnt.sum(0), 0) - cnt.sum(0))
    # received rows are labelled (s, e, i) in source-major order; expert-major order sorts by (e, s, i)
    labels = [(s, e, i) for s in range(G) for e in range(EL) for i in range(int(cnt[s, e]))]
    want = sorted(range(R), key=lambda r: (labels[r][1], labels[r][0], labels[r][2]))
    assert gather.cpu().tolist() == want
    assert scatter.cpu().long()[gather.cpu().long()].tolist() == list(range(R))


def test_expert_parallel_wrapper_device_path_world1(fq):
    """ExpertParallelMoE at world size 1 on the GPU (plan + fused gather + grouped GEMM + fused combine) equals
    dispatch -> grouped GEMM -> un-sort -> weighted sum written with torch ops."""
    from fused_int4_amd import ops, routing as R
    from fused_int4_amd.ep import ExpertParallelMoE
    E, N, K, T, top_k = 4, 136, 256, 50, 2
    P, S, Z, _, _, _ = make_moe(E, N, K, [1] * E, 21)
    route = R.simulate_routing(T, E, top_k, "skewed", "cuda", 7)
    x = torch.randn(T, K, generator=torch.Generator().manual_seed(1)).cuda()
    ep = ExpertParallelMoE(E, dev(P), dev(S), dev(Z))
    got = ep(x, route.expert_indices, route.expert_weights)
    flat = route.expert_indices.reshape(-1)
    order = torch.argsort(flat, stable=True)
    rows = x.index_select(0, order // top_k)
    tpe = torch.bincount(flat, minlength=E).to(torch.int32)
    offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
    y = ops.moe_forward(dev(P), dev(S), dev(Z), rows, None, tpe, offs)
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(order.numel(), device=order.device)
    want = (y.index_select(0, inverse).view(T, top_k, N) * route.expert_weights.unsqueeze(-1)).sum(dim=1)
    assert torch.equal(got, want)


@pytest.mark.parametrize("prec,tol", [("exact", 2e-5), ("fast", 1e-3)])
def test_gated_ffn_experts(fq, prec, tol):
    """down(silu(gate(x)) * up(x)) per expert; gate|up as one grouped GEMM, the activation fused into
    the down GEMM's pre-pass.  Against the float64 oracle on the same quantised weights.  (Two chained GEMMs and an
    expf: the stated bound is 2e-5 relative in exact mode; the intermediate [T, 2F] is float32.)"""
    torch.manual_seed(13)
    E, H, F = 3, 128, 96
    gate = [torch.randn(F, H) * 0.08 for _ in range(E)]
    up = [torch.randn(F, H) * 0.08 for _ in range(E)]
    down = [torch.randn(H, F) * 0.08 for _ in range(E)]
    ffn = fq.QuantizedMoEFFN.from_weights(gate, up, down, precision=prec).cuda()
    counts = np.array([37, 0, 70], dtype=np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    x = torch.randn(int(counts.sum()), H)
    out = ffn(x.cuda(), dev(counts), dev(offs)).cpu().numpy()
    ref = O.gated_ffn_grouped(
        tuple(t.cpu().numpy() for t in (ffn.gate_up_packed, ffn.gate_up_scales, ffn.gate_up_zero_points)),
        tuple(t.cpu().numpy() for t in (ffn.down_packed, ffn.down_scales, ffn.down_zero_points)),
        x.numpy(), counts, offs)
    assert out.shape == ref.shape
    assert rel_fro(out, ref) < tol
    # the fused activation equals the un-fused composition (same kernels, activation by torch) to float32 rounding
    from fused_int4_amd import ops
    gu = ops.moe_forward(ffn.gate_up_packed, ffn.gate_up_scales, ffn.gate_up_zero_points, x.cuda(), None, dev(counts),
                         dev(offs), precision=prec)
    h = torch.nn.functional.silu(gu[:, :F]) * gu[:, F:]
    unfused = ops.moe_forward(ffn.down_packed, ffn.down_scales, ffn.down_zero_points, h.contiguous(), None, dev(counts),
                              dev(offs), precision=prec).cpu().numpy()
    assert rel_fro(out, unfused) < (1e-5 if prec == "exact" else 5e-4)


def test_moe_clipping_of_bad_ranges(fq):
    """Offsets / counts that leave [0, T] are clipped on the device; nothing faults."""
    from fused_int4_amd import ops
    P, S, Z, x, cnt, offs = make_moe(3, 128, 256, [10, 10, 10], 5)
    cnt2 = np.array([10, 10, 999], np.int32)
    offs2 = np.array([0, 10, 20], np.int32)
    out = ops.moe_forward(dev(P), dev(S), dev(Z), dev(x), None, dev(cnt2), dev(offs2)).cpu().numpy()
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert rel_fro(out, ref) < EXACT_REL_FRO


# ------------------------------------------------------------------------------ BASELINE.json sizes: size-independent properties
def test_full_size_moe_properties(fq):
    """configs[2]: 8 experts 4096 -> 11008, 1024 routed rows.  Checked by (a) the oracle on a row sample,
    (b) linearity in the activations (exact for power-of-two scaling), (c) grouped == per-expert
    bit for bit, (d) column-checksum against the dequantised-weight checksum."""
    from fused_int4_amd import ops
    E, K, N = 8, 4096, 11008
    g = torch.Generator(device="cuda").manual_seed(9)
    P, S, Z = [], [], []
    for _ in range(E):
        p, s, z = fq.quantize_weights(torch.randn(N, K, device="cuda", generator=g) * 0.02)
        P.append(p); S.append(s); Z.append(z)
    P, S, Z = torch.stack(P), torch.stack(S), torch.stack(Z)
    cnt = torch.tensor([128] * 8, dtype=torch.int32, device="cuda")
    offs = torch.cumsum(cnt, 0).to(torch.int32) - cnt
    x = torch.randn(1024, K, device="cuda", generator=g)
    out = ops.moe_forward(P, S, Z, x, None, cnt, offs)
    # (a) oracle on sampled rows (float64 accumulation on the CPU)
    rows = [0, 127, 128, 500, 1023]
    Pn, Sn, Zn = P.cpu().numpy(), S.cpu().numpy(), Z.cpu().numpy()
    for r in rows:
        e = r // 128
        ref = C.linear_f64acc(x[r].cpu().numpy(), Pn[e], Sn[e], Zn[e])
        assert rel_fro(out[r].cpu().numpy(), ref) < EXACT_REL_FRO, r
        assert np.allclose(out[r].cpu().numpy(), ref, atol=1e-3)
    # (b) scaling the activations by 2^k scales the outputs exactly
    out4 = ops.moe_forward(P, S, Z, x * 4.0, None, cnt, offs)
    assert torch.equal(out4, out * 4.0)
    # (c) grouped launch == single-expert launch on the same rows, bit for bit
    single = ops.linear_forward(x[256:384].contiguous(), P[2], S[2], Z[2])
    assert torch.equal(single, out[256:384])
    # (c') the balanced column tiling (64 tiles of 192 / 160 columns per row block at this shape) returns the bits of
    #      the plain one (58 tiles of 192), for even routing and for a ragged one where the kernel itself falls back
    #      to the plain tiling
    from fused_int4_amd import _native
    lib = _native.lib()
    cnt2 = torch.tensor([485, 312, 126, 48, 30, 13, 6, 4], dtype=torch.int32, device="cuda")
    offs2 = torch.cumsum(cnt2, 0).to(torch.int32) - cnt2
    out_r = ops.moe_forward(P, S, Z, x, None, cnt2, offs2)
    try:
        assert lib.fql_tune_set_balance_tiles(0) == 1
        assert torch.equal(ops.moe_forward(P, S, Z, x, None, cnt, offs), out)
        assert torch.equal(ops.moe_forward(P, S, Z, x, None, cnt2, offs2), out_r)
    finally:
        lib.fql_tune_set_balance_tiles(1)
    r0 = 485 + 312 + 126 + 48 + 30 + 13                       # the two rows of the 6-row expert and the 4-row expert's first
    for r, e in ((r0, 6), (r0 + 5, 6), (r0 + 6, 7)):
        ref = C.linear_f64acc(x[r].cpu().numpy(), Pn[e], Sn[e], Zn[e])
        assert rel_fro(out_r[r].cpu().numpy(), ref) < EXACT_REL_FRO, r
    # (d) checksum of checksums: sum_n out[t,n] == x[t] . (sum_n W[n,:])
    wsum = fq.dequantize_weights(P[5], S[5], Z[5]).double().sum(0)
    lhs = out[640:768].double().sum(1)
    rhs = x[640:768].double() @ wsum
    assert torch.allclose(lhs, rhs, rtol=1e-5, atol=1e-3)


def test_full_size_linear_b512(fq):
    """configs[1]: QuantizedLinear 4096 -> 11008, batch 512."""
    from fused_int4_amd import ops
    K, N = 4096, 11008
    g = torch.Generator(device="cuda").manual_seed(10)
    p, s, z = fq.quantize_weights(torch.randn(N, K, device="cuda", generator=g) * 0.02)
    x = torch.randn(512, K, device="cuda", generator=g)
    out = ops.linear_forward(x, p, s, z)
    pn, sn, zn = p.cpu().numpy(), s.cpu().numpy(), z.cpu().numpy()
    for r in (0, 255, 511):
        ref = C.linear_f64acc(x[r].cpu().numpy(), pn, sn, zn)
        assert rel_fro(out[r].cpu().numpy(), ref) < EXACT_REL_FRO
    # batch-1 GEMV kernel (float32 FMA) agrees with the MFMA kernel to float32 noise
    o1 = ops.linear_forward(x[7], p, s, z)
    assert rel_fro(o1.cpu().numpy(), out[7].cpu().numpy()) < 5e-6
    # fast mode stays inside the north-star bound (<= 1e-3 relative) by a wide margin
    of = ops.linear_forward(x, p, s, z, precision="fast")
    assert rel_fro(of[:16].cpu().numpy(), out[:16].cpu().numpy()) < FAST_REL_FRO


def test_balanced_column_tiles_ragged_n(fq):
    """Uneven column tiles (fql_gemm_i8.h, tile_params) on a shape where the balanced tile count differs from the plain
    one AND N is not a multiple of the fragment width: 16 row blocks x 170 fragments (the last one 29 columns wide) =
    464 plain tiles on 256 CUs -> 32 tiles of 6 / 5 fragments per block.  Against the float64 oracle, and bit for bit
    against the plain tiling."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    B, N, K = 2048, 5437, 256
    x, p, s, z = make_problem(N, K, B, 5437)
    dx, dp, ds, dz = dev(x), dev(p), dev(s), dev(z)
    out = ops.linear_forward(dx, dp, ds, dz)
    ref = C.linear_f64acc(x, p, s, z)
    assert rel_fro(out.cpu().numpy(), ref) < EXACT_REL_FRO
    try:
        assert lib.fql_tune_set_balance_tiles(0) == 1
        plain = ops.linear_forward(dx, dp, ds, dz)
    finally:
        lib.fql_tune_set_balance_tiles(1)
    assert torch.equal(out, plain)
    for prec in ("fast", "int8"):
        o = ops.linear_forward(dx, dp, ds, dz, precision=prec)
        try:
            lib.fql_tune_set_balance_tiles(0)
            assert torch.equal(o, ops.linear_forward(dx, dp, ds, dz, precision=prec)), prec
        finally:
            lib.fql_tune_set_balance_tiles(1)


# ------------------------------------------------------------------------------ per-group scales along K
@pytest.mark.parametrize("B,N,K,group", [(1, 64, 256, 64), (7, 200, 512, 128), (40, 96, 1024, 32), (3, 33, 96, 2),
                                         (200, 100, 192, 64), (300, 130, 512, 128), (130, 64, 256, 32), (5, 40, 64, 32),
                                         (16, 72, 128, 16), (2, 100, 512, 128), (3, 72, 256, 32)])
def test_per_group_scales_linear(fq, B, N, K, group):
    """Not in the reference (per-row only): checked against the float64 dequantize-then-matmul with per-group constants.
    Shapes cover the one-wave-per-row kernel (odd groups), the GEMV kernel with per-group constants (<= 3 rows) and both forms of the float32 matrix-core kernel
    (csrc/fql_group.h: 32 x 32 blocks with K split over the waves, 64 x 64 tiles), ragged in rows and columns."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(B + K)
    w = rng.standard_normal((N, K)).astype(np.float32)
    p, s, z = O.quantize_weights_grouped(w, group)
    x = rng.standard_normal((B, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    got = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    ref = O.reference_linear_grouped(x, p, s, z)
    assert rel_fro(got, ref) < FMA_REL_FRO
    gotb = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), bias=dev(bias)).cpu().numpy()
    assert np.array_equal(gotb, got + bias[None, :])
    # GPU quantiser + module
    m = fq.QuantizedLinear.from_linear(torch.nn.Linear(K, N, bias=False), group_size=group).cuda()
    assert m.scales.shape == (N, K // group)
    xt = torch.from_numpy(x)
    assert torch.allclose(m(xt.cuda()).cpu(), m.cpu()(xt), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("B,N,K,group", [(64, 256, 1024, 128), (512, 384, 4096, 128), (50, 100, 512, 64), (49, 72, 512, 256)])
@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO_LARGE_K)])
def test_per_group_scales_integer_matrix_cores(fq, B, N, K, group, prec, tol):
    """Batches (>= 40 rows of one matrix, >= 8 rows per expert) with K % 256 == 0 and group % 64 == 0: the per-group path on the INT8 matrix cores
    (csrc/fql_group_i8.h: per-group integer dot products and limb sums, folded in float32 at the end of every group),
    against the float64 per-group dequantize-then-matmul, at the tolerances of the per-row modes; and the float32
    matrix-core path (the same call with the integer kernel switched off) agrees with it to float32 noise."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    rng = np.random.default_rng(B + K + group)
    w = rng.standard_normal((N, K)).astype(np.float32)
    p, s, z = O.quantize_weights_grouped(w, group)
    x = rng.standard_normal((B, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = O.reference_linear_grouped(x, p, s, z)
    got = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec).cpu().numpy()
    assert rel_fro(got, ref) < tol, rel_fro(got, ref)
    gotb = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec, bias=dev(bias)).cpu().numpy()
    assert np.array_equal(gotb, got + bias[None, :])
    if prec == "exact":
        try:
            assert lib.fql_tune_set_group_i8(0) == 1
            f32 = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
        finally:
            lib.fql_tune_set_group_i8(1)
        assert rel_fro(f32, ref) < FMA_REL_FRO
        assert rel_fro(got, f32) < 3e-6 and not np.array_equal(got, f32)      # two different kernels did run


@pytest.mark.parametrize("E,N,K,group,counts", [(4, 96, 256, 64, [9, 0, 17, 5]), (4, 136, 512, 128, [150, 0, 40, 3]), (3, 136, 512, 128, [150, 70, 30]),
                                                (3, 70, 192, 32, [260, 1, 66])])
def test_per_group_scales_grouped_moe(fq, E, N, K, group, counts):
    from fused_int4_amd import ops
    rng = np.random.default_rng(5)
    counts = np.array(counts, np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    T = int(counts.sum()) + 2
    q = [O.quantize_weights_grouped(rng.standard_normal((N, K)).astype(np.float32), group) for _ in range(E)]
    P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
    x = rng.standard_normal((T, K)).astype(np.float32)
    out = ops.moe_group_forward(dev(P), dev(S), dev(Z), dev(x), dev(counts), dev(offs)).cpu().numpy()
    for e in range(E):
        o, c = int(offs[e]), int(counts[e])
        if c:
            assert rel_fro(out[o:o + c], O.reference_linear_grouped(x[o:o + c], P[e], S[e], Z[e])) < FMA_REL_FRO
    assert (out[T - 2:] == 0).all()
