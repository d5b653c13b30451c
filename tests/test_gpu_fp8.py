"""fp8 (OCP e4m3) activations x INT4 weights on the block-scaled fp8 matrix-core instruction
(v_mfma_scale_f32_32x32x64_f8f6f4), BASELINE.json configs[4].

The reference has no fp8 path (README.md:228 lists it as future work), so there is no reference vector: "parity
unpinned" by the reference.  The checker is the reference's dequantize-then-matmul (python/quantize.py:176-202)
applied in float64 to the decoded e4m3 activations (oracle.reference_linear_fp8); the e4m3 format itself is pinned to
torch's float8_e4m3fn casts (tests/test_oracle_golden.py).  Two tolerances, both in tests/helpers.py:
FP8_ACC_REL_FRO (kernel vs float64 on the SAME fp8 inputs: accumulation only) and FP8_FORMAT_REL_FRO (vs float32
activations: the format's own rounding)."""
import numpy as np
import pytest
import torch

from helpers import FP8_ACC_REL_FRO, FP8_ACC_SCALED_REL_FRO, FP8_FORMAT_REL_FRO, rel_fro, decode_limbs
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


def random_e4m3(rng, shape):
    b = rng.integers(0, 256, size=shape, dtype=np.uint8)
    b[(b & 0x7F) == 0x7F] ^= 1                              # no NaN
    return b


def test_small_integers_are_exact(fq):
    """Integer activations in [-8, 8] are exact in e4m3, the 4-bit weights are exact in e4m3 (as q * 2^-9 with a 2^9
    block scale), every partial sum is an integer below 2^24: the fp8 pass must return the integer dot product exactly."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(1)
    B, N, K = 70, 320, 1024
    q = rng.integers(0, 16, size=(N, K), dtype=np.uint8)
    p = ((q[:, 1::2] << 4) | q[:, 0::2]).astype(np.uint8)
    zp = rng.integers(0, 16, size=N).astype(np.float32)
    sc = (rng.random(N).astype(np.float32) + 0.5) * np.float32(0.01)
    xi = rng.integers(-8, 9, size=(B, K))
    x8 = O.e4m3_encode(xi.astype(np.float32))
    assert np.array_equal(O.e4m3_decode(x8), xi.astype(np.float32))
    ei = xi.astype(np.int64) @ (q.astype(np.int64) - zp.astype(np.int64)[:, None]).T
    want = (ei.astype(np.float32) * sc[None, :]).astype(np.float32)
    got = ops.linear_forward_fp8(dev(x8), None, dev(p), dev(sc), dev(zp)).cpu().numpy()
    assert np.array_equal(got, want)
    # torch.float8_e4m3fn tensors are taken as they are
    got2 = ops.linear_forward_fp8(dev(x8).view(torch.float8_e4m3fn), None, dev(p), dev(sc), dev(zp)).cpu().numpy()
    assert np.array_equal(got2, want)


@pytest.mark.parametrize("B,N,K", [(1, 64, 32), (4, 200, 96), (5, 64, 128), (33, 200, 544), (100, 1000, 1024), (64, 192, 4096),
                                   (129, 193, 7168), (300, 400, 320)])
def test_fp8_rows_against_float64(fq, B, N, K):
    """Any finite e4m3 bytes with per-row scales; ragged M, N, K tails; rows longer than 4096."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(B + K)
    p, s, z = O.quantize_weights(rng.standard_normal((N, K)).astype(np.float32))
    x8 = random_e4m3(rng, (B, K))
    sc = (rng.random(B).astype(np.float32) + 0.25)
    got = ops.linear_forward_fp8(dev(x8), dev(sc), dev(p), dev(s), dev(z)).cpu().numpy()
    ref = O.reference_linear_fp8(x8, sc, p, s, z)
    assert got.shape == ref.shape
    assert rel_fro(got, ref) < FP8_ACC_REL_FRO, rel_fro(got, ref)
    got1 = ops.linear_forward_fp8(dev(x8), None, dev(p), dev(s), dev(z)).cpu().numpy()
    assert rel_fro(got1, O.reference_linear_fp8(x8, None, p, s, z)) < FP8_ACC_REL_FRO


def test_fused_fp8_quantiser_is_bit_exact(fq):
    """precision="fp8" pre-pass: per-row scale max|x| / 448, e4m3 round-to-nearest-even of the float32 quotient --
    bytes, scales and row sums against the numpy restatement (whose rounding is pinned to torch's cast)."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(8)
    T, K = 21, 4128                                         # multi-slab rows, K % 256 != 0
    x = rng.standard_normal((T, K)).astype(np.float32)
    x[3] = 0.0
    x[4] *= 1e-20
    x[5] *= 1e20
    x[6, 17] = 3.0e4                                        # an outlier: the rest of the row lands in the subnormals
    x[7] = np.float32(448.0)
    limbs, delta, rowsum = ops.act_quant(dev(x), precision="fp8")
    Kp = (K + 255) // 256 * 256
    dig, covered = decode_limbs(limbs.cpu().numpy(), 1, T, 1, K, Kp)
    got_bytes = (dig[0].astype(np.int64) & 0xFF).astype(np.uint8)
    want_bytes, want_scale = O.quantize_activations_fp8(x)
    assert covered.all()
    assert np.array_equal(delta[0].cpu().numpy(), want_scale)
    assert np.array_equal(got_bytes[:, :K] & 0x7F, want_bytes & 0x7F)                     # magnitudes
    nz = (want_bytes & 0x7F) != 0
    assert np.array_equal((got_bytes[:, :K] & 0x80)[nz], (want_bytes & 0x80)[nz])         # signs (of non-zeros)
    assert (got_bytes[:, K:] == 0).all()
    want_sum = (O.e4m3_decode(want_bytes).astype(np.float64).sum(axis=1)).astype(np.float32)
    assert np.array_equal(rowsum.cpu().numpy().view(np.float32)[0, 0], want_sum)


@pytest.mark.parametrize("B,N,K", [(5, 96, 64), (40, 256, 1024), (130, 192, 4096)])
def test_precision_fp8_end_to_end(fq, B, N, K):
    from fused_int4_amd import ops
    rng = np.random.default_rng(B * K)
    p, s, z = O.quantize_weights(rng.standard_normal((N, K)).astype(np.float32))
    x = rng.standard_normal((B, K)).astype(np.float32)
    got = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision="fp8").cpu().numpy()
    x8, sc = O.quantize_activations_fp8(x)
    assert rel_fro(got, O.reference_linear_fp8(x8, sc, p, s, z)) < FP8_ACC_SCALED_REL_FRO   # the kernel's own error
    err = rel_fro(got, C.linear_f64acc(x, p, s, z))                                       # the format's error
    assert 5e-3 < err < FP8_FORMAT_REL_FRO, err
    # the two entry paths agree bit for bit: fused quantiser == torch-side quantiser + fp8 entry point
    xq, xs = ops.quantize_activations_fp8(dev(x))
    assert np.array_equal(xq.view(torch.uint8).cpu().numpy() & 0x7F, x8 & 0x7F)
    got2 = ops.linear_forward_fp8(xq, xs, dev(p), dev(s), dev(z)).cpu().numpy()
    assert np.array_equal(got2, got)


def make_moe(E, N, K, counts, seed, gap_rows=0):
    rng = np.random.default_rng(seed)
    q = [O.quantize_weights((rng.standard_normal((N, K)) * 0.02).astype(np.float32)) for _ in range(E)]
    P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
    counts = np.asarray(counts, dtype=np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    T = int(counts.sum()) + gap_rows
    return P, S, Z, counts, offs, T, rng


@pytest.mark.parametrize("E,N,K,counts,gap", [
    (4, 256, 128, [40, 0, 33, 20], 3),
    (8, 384, 512, [128] * 8, 0),
    (8, 200, 256, [300, 1, 0, 17, 129, 64, 0, 5], 11),
    (70, 96, 128, [0] * 64 + [33, 0, 1, 40, 0, 2], 0),
    (6, 160, 7168, [48, 5, 0, 70, 16, 1], 2),
])
def test_moe_fp8_grouped(fq, E, N, K, counts, gap):
    from fused_int4_amd import ops
    P, S, Z, cnt, offs, T, rng = make_moe(E, N, K, counts, sum(counts) + E, gap)
    x8 = random_e4m3(rng, (T, K))
    sc = (rng.random(T).astype(np.float32) + 0.25)
    out = ops.moe_forward_fp8(dev(P), dev(S), dev(Z), dev(x8), dev(sc), dev(cnt), dev(offs)).cpu().numpy()
    ref = O.reference_moe_grouped_fp8(x8, sc, P, S, Z, cnt, offs)
    assert rel_fro(out, ref) < FP8_ACC_REL_FRO, rel_fro(out, ref)
    covered = np.zeros(T, bool)
    for c, o in zip(cnt, offs):
        covered[o:o + c] = True
    assert (out[~covered] == 0).all()
    # float rows through the same grouped op with precision="fp8"
    x = rng.standard_normal((T, K)).astype(np.float32)
    got = ops.moe_forward(dev(P), dev(S), dev(Z), dev(x), None, dev(cnt), dev(offs), precision="fp8").cpu().numpy()
    xq, xs = O.quantize_activations_fp8(x)
    assert rel_fro(got, O.reference_moe_grouped_fp8(xq, xs, P, S, Z, cnt, offs)) < FP8_ACC_SCALED_REL_FRO
    assert (got[~covered] == 0).all()
    # 16-bit outputs are the rounded float32 outputs
    o16 = ops.moe_forward_fp8(dev(P), dev(S), dev(Z), dev(x8), dev(sc), dev(cnt), dev(offs), out_dtype=torch.bfloat16)
    assert torch.equal(o16.cpu(), torch.from_numpy(out).to(torch.bfloat16))


def test_fp8_tile_configurations_agree(fq):
    """Every tile shape runs the same per-output instruction sequence over K, so the float32 results are bit-identical."""
    from fused_int4_amd import ops
    P, S, Z, cnt, offs, T, rng = make_moe(5, 200, 768, [0, 7, 33, 70, 129], 123)
    x = rng.standard_normal((T, 768)).astype(np.float32)
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision="fp8", tokens_per_expert=dc, input_offsets=do)
    outs = {}
    for cfg in (1, 5, 6, 7, 8, 11, 12):
        out = torch.full((T, 200), float("nan"), dtype=torch.float32, device="cuda")
        rc = ops.tune_gemm_i8(cfg, limbs, delta, rowsum, dP, dS, dZ, dc, do, out, 5, T, 768, 200, "fp8")
        assert rc == 0, (cfg, rc)
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy()
    xq, xs = O.quantize_activations_fp8(x)
    assert rel_fro(outs[1], O.reference_moe_grouped_fp8(xq, xs, P, S, Z, cnt, offs)) < FP8_ACC_SCALED_REL_FRO
    for cfg, o in outs.items():
        assert np.array_equal(o, outs[1]), cfg


def test_fp8_nan_and_errors(fq):
    from fused_int4_amd import ops
    rng = np.random.default_rng(3)
    p, s, z = O.quantize_weights(rng.standard_normal((64, 128)).astype(np.float32))
    x8 = random_e4m3(rng, (9, 128))
    x8[4, 77] = 0x7F
    out = ops.linear_forward_fp8(dev(x8), None, dev(p), dev(s), dev(z)).cpu().numpy()
    assert np.isnan(out[4]).all() and np.isfinite(np.delete(out, 4, axis=0)).all()
    with pytest.raises(RuntimeError, match="float8_e4m3fn"):
        ops.linear_forward_fp8(dev(x8).float(), None, dev(p), dev(s), dev(z))
    with pytest.raises(RuntimeError, match="% 32"):
        ops.linear_forward_fp8(dev(x8[:, :34]), None, dev(p[:, :17]), dev(s), dev(z))
    with pytest.raises(RuntimeError, match="one element per row"):
        ops.linear_forward_fp8(dev(x8), dev(np.ones(3, np.float32)), dev(p), dev(s), dev(z))
    # float rows, K % 32 != 0: the fp8 arithmetic exists on the MFMA path only -> an error, never a silent float path
    x = rng.standard_normal((8, 34)).astype(np.float32)
    p2, s2, z2 = O.quantize_weights(rng.standard_normal((16, 34)).astype(np.float32))
    with pytest.raises(RuntimeError):
        ops.linear_forward(dev(x), dev(p2), dev(s2), dev(z2), precision="fp8")
