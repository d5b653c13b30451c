"""GPU parity beyond K = 4096 and at BASELINE.json configs[4] (64 experts 7168 -> 18432, top-6).

Rows longer than 4096 take the activation pre-pass's multi-slab branch (csrc/fql_act_quant.h: pass 1 streams the
slabs for the row maximum, pass 2 re-reads them) and GEMMs with more than 16 weight stages; QuantizedMoEFFN
down-projections (K = 11008 / 14336) and configs[4] (K = 7168) are such shapes.  Checker: the float64-accumulating
C oracle (oracle/int4_oracle.c) on the same quantised weights, tolerances of tests/helpers.py.

Reference for the shapes: benchmark/moe_grouped_gemm/config.py:78-84 (DeepSeek-style configuration) and
BASELINE.json configs[4]; for the arithmetic: python/quantize.py:127-202."""
import numpy as np
import pytest
import torch

from helpers import (EXACT_REL_FRO, FAST_REL_FRO, INT8_REL_FRO_LARGE_K as INT8_REL_FRO, FP8_ACC_SCALED_REL_FRO, rel_fro, act_limbs_reference,
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
    _native.lib()
    return pkg


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def quantised(rng, N, K, wscale=0.02):
    return O.quantize_weights((rng.standard_normal((N, K)) * wscale).astype(np.float32))


@pytest.mark.parametrize("L,prec", [(3, "exact"), (2, "fast"), (1, "int8")])
@pytest.mark.parametrize("K", [4128, 7168])
def test_activation_limbs_bit_exact_multi_slab(fq, L, prec, K):
    """Rows longer than one 4096-k slab: limbs, delta and row sums against the numpy restatement of the rule."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(K + L)
    T = 11
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[2, K - 1] = 37.5                                      # the row maximum sits in the LAST slab
    x[3, 4096] = -91.0                                      # ... and at the first element of the second slab
    x[4] = 0.0
    limbs, delta, rowsum = ops.act_quant(dev(x), precision=prec)
    Kp = (K + 255) // 256 * 256
    dig, covered = decode_limbs(limbs.cpu().numpy(), L, T, 1, K, Kp)
    ref_dig, ref_delta, ref_sum = act_limbs_reference(x, L)
    assert covered.all()
    assert np.array_equal(delta[0].cpu().numpy(), ref_delta)
    assert np.array_equal(dig[:, :, :K], ref_dig)
    assert (dig[:, :, K:] == 0).all()
    assert np.array_equal(rowsum[0].cpu().numpy(), ref_sum)


@pytest.mark.parametrize("B,N,K", [(40, 192, 4128), (37, 256, 7168), (130, 200, 11008), (16, 128, 14336)])
@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO)])
def test_dense_large_k(fq, B, N, K, prec, tol):
    from fused_int4_amd import ops
    rng = np.random.default_rng(B + K)
    p, s, z = quantised(rng, N, K, 1.0)
    x = rng.standard_normal((B, K)).astype(np.float32)
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec).cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    assert rel_fro(out, ref) < tol
    if prec == "exact":                                     # the reference's own GPU tolerance at its largest K
        assert np.allclose(out, ref, atol=1e-2, rtol=1e-5)  # tests/test_correctness.py:252


@pytest.mark.parametrize("K,counts", [(4128, [33, 0, 130, 7]), (7168, [48, 48, 1, 70]), (11008, [16, 5, 0, 40])])
@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", FAST_REL_FRO), ("int8", INT8_REL_FRO)])
def test_grouped_large_k(fq, K, counts, prec, tol):
    from fused_int4_amd import ops
    rng = np.random.default_rng(K)
    E, N = len(counts), 160
    q = [quantised(rng, N, K) for _ in range(E)]
    P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
    cnt = np.asarray(counts, np.int32)
    offs = (np.cumsum(cnt) - cnt).astype(np.int32)
    x = rng.standard_normal((int(cnt.sum()), K)).astype(np.float32)
    out = ops.moe_forward(dev(P), dev(S), dev(Z), dev(x), None, dev(cnt), dev(offs), precision=prec).cpu().numpy()
    assert rel_fro(out, C.moe_grouped(P, S, Z, x, cnt, offs)) < tol


@pytest.mark.parametrize("prec,tol", [("exact", 2e-5), ("fast", 1e-3)])
def test_gated_ffn_large_hidden(fq, prec, tol):
    """Down projection with K = F = 4352 > 4096: the GATE variant of the multi-slab pre-pass."""
    torch.manual_seed(3)
    E, H, F = 2, 64, 4352
    gate = [torch.randn(F, H) * 0.1 for _ in range(E)]
    up = [torch.randn(F, H) * 0.1 for _ in range(E)]
    down = [torch.randn(H, F) * 0.05 for _ in range(E)]
    ffn = fq.QuantizedMoEFFN.from_weights(gate, up, down, precision=prec).cuda()
    counts = np.array([21, 9], dtype=np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    x = torch.randn(int(counts.sum()), H)
    out = ffn(x.cuda(), dev(counts), dev(offs)).cpu().numpy()
    ref = O.gated_ffn_grouped(
        tuple(t.cpu().numpy() for t in (ffn.gate_up_packed, ffn.gate_up_scales, ffn.gate_up_zero_points)),
        tuple(t.cpu().numpy() for t in (ffn.down_packed, ffn.down_scales, ffn.down_zero_points)),
        x.numpy(), counts, offs)
    assert rel_fro(out, ref) < tol


# ------------------------------------------------------------------------------ BASELINE.json configs[4], full size
def _config5_weights(fq, E, N, K, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    P = torch.empty((E, N, K // 2), dtype=torch.uint8, device="cuda")
    S = torch.empty((E, N), dtype=torch.float32, device="cuda")
    Z = torch.empty((E, N), dtype=torch.float32, device="cuda")
    for e in range(E):
        w = torch.randn(N, K, device="cuda", generator=g) * 0.02
        P[e], S[e], Z[e] = fq.quantize_weights(w)
        del w
    return P, S, Z


@pytest.fixture(scope="module")
def config5(fq):
    """64 experts 7168 -> 18432, 512 tokens top-6 = 3072 routed rows: balanced (48 per expert) and ragged counts."""
    E, K, N = 64, 7168, 18432
    P, S, Z = _config5_weights(fq, E, N, K, 17)
    g = torch.Generator(device="cuda").manual_seed(18)
    x = torch.randn(3072, K, device="cuda", generator=g)
    yield E, K, N, P, S, Z, x
    del P, S, Z, x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("int8", INT8_REL_FRO), ("fp8", FP8_ACC_SCALED_REL_FRO)])
@pytest.mark.parametrize("routing", ["balanced", "ragged"])
def test_full_size_config5(fq, config5, prec, tol, routing):
    from fused_int4_amd import ops
    E, K, N, P, S, Z, x = config5
    if routing == "balanced":
        cnt_h = np.full(E, 48, np.int32)
    else:
        rng = np.random.default_rng(5)
        cnt_h = rng.multinomial(3072 - 200, np.ones(E) / E).astype(np.int32)
        cnt_h[7] += 200                                     # one hot expert, and
        cnt_h[9] = 0                                        # an empty one
        cnt_h[-1] += 3072 - int(cnt_h.sum())
    offs_h = (np.cumsum(cnt_h) - cnt_h).astype(np.int32)
    cnt, offs = dev(cnt_h), dev(offs_h)
    out = ops.moe_forward(P, S, Z, x, None, cnt, offs, precision=prec)
    assert out.shape == (3072, N)
    # (a) the oracle on the first / last / an inner row of several experts
    for e in (0, 7, 31, 63):
        c, o = int(cnt_h[e]), int(offs_h[e])
        if c == 0:
            continue
        pe, se, ze = P[e].cpu().numpy(), S[e].cpu().numpy(), Z[e].cpu().numpy()
        for r in sorted({o, o + c // 2, o + c - 1}):
            xr = x[r].cpu().numpy()
            if prec == "fp8":       # the oracle on the SAME e4m3 activations (the config's own format): kernel error only
                xq, xs = O.quantize_activations_fp8(xr[None])
                ref = C.linear_f64acc(O.e4m3_decode(xq[0]), pe, se, ze) * np.float64(xs[0])
            else:
                ref = C.linear_f64acc(xr, pe, se, ze)
            got = out[r].cpu().numpy()
            assert rel_fro(got, ref) < tol, (e, r)
            if prec == "exact":
                assert np.allclose(got, ref, atol=1e-2, rtol=1e-5), (e, r)
    # (b) scaling the activations by 2^k scales every output exactly (power-of-two row scales)
    out8 = ops.moe_forward(P, S, Z, x * 8.0, None, cnt, offs, precision=prec)
    assert torch.equal(out8, out * 8.0)
    del out8
    # (c) grouped launch == single-expert launch on the same rows, bit for bit (other tiles, other row offsets)
    for e in (7, 40):
        c, o = int(cnt_h[e]), int(offs_h[e])
        if c > 4:
            single = ops.linear_forward(x[o:o + c].contiguous(), P[e], S[e], Z[e], precision=prec)
            assert torch.equal(single, out[o:o + c]), e
    # (d) checksum of checksums on one expert
    e = 31
    c, o = int(cnt_h[e]), int(offs_h[e])
    wsum = fq.dequantize_weights(P[e], S[e], Z[e]).double().sum(0)
    lhs = out[o:o + c].double().sum(1)
    rhs = x[o:o + c].double() @ wsum
    if prec == "exact":
        assert torch.allclose(lhs, rhs, rtol=1e-4, atol=1e-2)
    else:                                                   # 8-bit activations: the format's own rounding
        assert float((lhs - rhs).norm() / rhs.norm()) < 0.1
