"""Activation dynamic range: outlier channels and outlier rows (what LLM activations look like) through the
fixed-point MFMA path, against the float64-accumulating oracle.

The limbs are PER-ROW block fixed point (csrc/fql_act_quant.h): one power-of-two quantum delta[t] per row, set by
the row's largest magnitude, so one outlier element coarsens the quantum of its whole row.  Every element is
rounded once, by at most delta/2; for errors uncorrelated with the weights the output error of row t would be

    ||out_t - ref_t|| / ||ref_t||  ~=  delta[t] / sqrt(12) * sqrt(K) / ||x_t||_2        (tests/helpers.row_quantum_bound)

up to 2^-(8L-2) * sqrt(K / 12) (4.4e-6 at K = 4096 for 3 limbs: above the mode's stated 2e-6).  The pre-pass therefore
flags the rows whose prediction exceeds 1e-6 (3 limbs) / 2.5e-4 (2 limbs) and gives them a second limb set holding
the rounding residual, which the GEMM adds in a second pass over the tiles that contain such rows: 16L-2 bits for
those rows, nothing changes for the others.  The tests check the stated constants of tests/helpers.py on every case,
the residual limbs themselves bit for bit, and that a row's result does not depend on its neighbours.
Reference tolerances: tests/test_correctness.py:218,233,252."""
import numpy as np
import pytest
import torch

from helpers import (EXACT_REL_FRO, FAST_REL_FRO, rel_fro, row_quantum_bound, act_limbs_reference, act_residual_reference,
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


K, N, B = 4096, 512, 48


@pytest.fixture(scope="module")
def weights():
    rng = np.random.default_rng(2024)
    return O.quantize_weights(rng.standard_normal((N, K)).astype(np.float32))


def heavy_tailed(kind, factor, rng):
    x = rng.standard_normal((B, K)).astype(np.float32)
    if kind == "one_channel":
        x[:, 1234] *= factor
    elif kind == "four_channels":
        x[:, [5, 1234, 2049, 4095]] *= factor
    elif kind == "row_mixture":                             # every row its own outlier columns and magnitudes
        for t in range(B):
            cols = rng.choice(K, size=1 + t % 4, replace=False)
            x[t, cols] *= factor * (0.25 + rng.random())
    elif kind == "massive_rows":                            # a few rows scaled as a whole: harmless (per-row scale)
        x[::7] *= factor
    return x


CASES = [(k, f) for k in ("one_channel", "four_channels", "row_mixture") for f in (30.0, 100.0, 1000.0)] + [("massive_rows", 1000.0)]


@pytest.mark.parametrize("kind,factor", CASES)
def test_exact_mode_heavy_tails(fq, weights, kind, factor):
    from fused_int4_amd import ops
    p, s, z = weights
    x = heavy_tailed(kind, factor, np.random.default_rng(int(factor) + len(kind)))
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision="exact").cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    # the stated constant holds for every case of this sweep, row by row
    for t in range(B):
        assert rel_fro(out[t], ref[t]) < EXACT_REL_FRO, (t, rel_fro(out[t], ref[t]))
    assert rel_fro(out, ref) < EXACT_REL_FRO, rel_fro(out, ref)
    # the reference's own tolerance, scaled with the output magnitude where the outliers blow the outputs up (the
    # float32 FMA chain of the reference kernel, csrc/quantized_linear_kernel.cu:240-244, needs the same allowance)
    f32 = C.linear_fma(x, p, s, z)
    scale = max(1.0, float(np.abs(ref).max()) / 300.0)
    assert np.allclose(out, ref, atol=1e-2 * scale, rtol=1e-5), np.abs(out - ref).max()
    # at least as accurate as that float32 FMA chain
    assert rel_fro(out, ref) < rel_fro(f32, ref) + 3e-7, (rel_fro(out, ref), rel_fro(f32, ref))


@pytest.mark.parametrize("kind,factor", CASES)
def test_fast_mode_heavy_tails(fq, weights, kind, factor):
    """2 limbs (15-bit fixed point per row): inside the north-star 1e-3 bound on every case of the sweep."""
    from fused_int4_amd import ops
    p, s, z = weights
    x = heavy_tailed(kind, factor, np.random.default_rng(int(factor) + len(kind)))
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision="fast").cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    # a quarter of the north-star bound, row by row: rows predicted above 2.5e-4 carry the residual limb set
    for t in range(B):
        assert rel_fro(out[t], ref[t]) < 3e-4, (t, rel_fro(out[t], ref[t]), row_quantum_bound(x, 2)[t])
    assert rel_fro(out, ref) < 3e-4, rel_fro(out, ref)


def test_grouped_heavy_tails_are_per_row(fq, weights):
    """An outlier row must not change any other row's result (the scale is per row, not per tile)."""
    from fused_int4_amd import ops
    p, s, z = weights
    rng = np.random.default_rng(77)
    x = rng.standard_normal((B, K)).astype(np.float32)
    base = ops.linear_forward(dev(x), dev(p), dev(s), dev(z)).cpu().numpy()
    x2 = x.copy()
    x2[5, 100] = 3.0e4
    x2[40] *= 1e-3
    got = ops.linear_forward(dev(x2), dev(p), dev(s), dev(z)).cpu().numpy()
    keep = np.ones(B, bool)
    keep[[5, 40]] = False
    assert np.array_equal(got[keep], base[keep])


@pytest.mark.parametrize("L,prec", [(3, "exact"), (2, "fast")])
def test_residual_limbs_bit_exact(fq, L, prec):
    """The second limb set of flagged rows against the numpy restatement of the rule; unflagged rows have delta2 == 0."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(17 + L)
    T, Kx = 24, 1280
    x = rng.standard_normal((T, Kx)).astype(np.float32)
    x[1, 7] *= 1000.0                                       # clearly flagged
    x[2, [3, 900]] *= 300.0
    x[5] *= 1e-12
    x[5, 100] *= 5000.0
    x[9] = 0.0                                              # all-zero row: never flagged
    x[11, 5] = np.float32(2.0 ** 20)                        # one huge power of two
    limbs, delta, rowsum = ops.act_quant(dev(x), precision=prec)
    flag, rdig, delta2, rsum2 = act_residual_reference(x, L)
    got_flag = delta[1].cpu().numpy() != 0
    sure = np.ones(T, bool)                                 # rows whose flag does not hinge on the last float32 bits
    assert flag[[1, 2, 5, 11]].all() and not flag[[0, 9]].any()
    assert np.array_equal(got_flag[sure], flag[sure])
    assert np.array_equal(delta[1].cpu().numpy(), delta2)
    dig2, covered = decode_limbs(limbs.cpu().numpy(), L, T, 1, Kx, 1280, which=1)
    assert covered.all()
    assert np.array_equal(dig2[:, flag, :Kx], rdig[:, flag])
    assert np.array_equal(rowsum[1].cpu().numpy()[:, flag], rsum2[:, flag])
    # the main set is what it was
    dig, _ = decode_limbs(limbs.cpu().numpy(), L, T, 1, Kx, 1280, which=0)
    ref_dig, ref_delta, ref_sum = act_limbs_reference(x, L)
    assert np.array_equal(dig[:, :, :Kx], ref_dig) and np.array_equal(delta[0].cpu().numpy(), ref_delta)
    # main + residual reconstruct x to 16L-2 bits of the row maximum
    X = sum(ref_dig[l] * 256.0 ** l for l in range(L)) * ref_delta[:, None].astype(np.float64)
    R = sum(rdig[l] * 256.0 ** l for l in range(L)) * delta2[:, None].astype(np.float64)
    err = np.abs(x.astype(np.float64) - X - R)
    assert np.all(err[flag] <= 0.5 * delta2[flag, None] * (1 + 1e-7))


@pytest.mark.parametrize("prec", ["exact", "fast"])
def test_heavy_tails_every_tile_configuration(fq, prec):
    """Flagged and unflagged rows mixed in one ragged grouped problem: every tile configuration (wide, 32-row,
    16-row) returns the same bits, and equals the float64 oracle to the mode's constant."""
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
    for t in range(0, T, 5):                                # every fifth row heavy-tailed: some tiles mixed, some clean
        x[t, rng.choice(Kk, 2, replace=False)] *= 800.0
    x[40:60] *= 1.0                                         # a run of clean rows (a clean 16-row tile)
    x[40:56, :] = rng.standard_normal((16, Kk)).astype(np.float32)
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(counts), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision=prec, tokens_per_expert=dc, input_offsets=do)
    assert 0 < int((delta[1] != 0).sum()) < T
    outs = {}
    cfgs = [c for c in (list(range(lib.fql_tune_num_configs())) + list(range(100, 100 + lib.fql_tune_num_rows32_configs()))
                        + list(range(200, 200 + lib.fql_tune_num_rows16_configs()))
                        + list(range(300, 300 + lib.fql_tune_num_w4_configs())))
            if lib.fql_tune_is_config(c, ops._precision(prec))]      # (not every wide id is built for every limb count)
    assert len(cfgs) >= 20 and cfgs[0] in (0, 1)
    for cfg in cfgs:
        out = torch.full((T, Nn), float("nan"), dtype=torch.float32, device="cuda")
        rc = ops.tune_gemm_i8(cfg, limbs, delta, rowsum, dP, dS, dZ, dc, do, out, E, T, Kk, Nn, prec)
        assert rc == 0, (cfg, rc)
        torch.cuda.synchronize()
        outs[cfg] = out.cpu().numpy()
    ref = C.moe_grouped(P, S, Z, x, counts, offs)
    assert rel_fro(outs[cfgs[0]], ref) < (EXACT_REL_FRO if prec == "exact" else 3e-4)
    for cfg, o in outs.items():
        assert np.array_equal(o, outs[cfgs[0]]), f"configuration {cfg} differs from configuration {cfgs[0]}"
    # the product entry point (pre-pass + GEMM + scratch from the workspace) gives the same bits
    prod = ops.moe_forward(dP, dS, dZ, dx, None, dc, do, precision=prec).cpu().numpy()
    assert np.array_equal(prod, outs[cfgs[0]])
    # 16-bit outputs: rounded once, from the float32 sum of main and residual parts
    o16 = ops.moe_forward_any(dP, dS, dZ, dx, None, dc, do, precision=prec, out_dtype=torch.bfloat16)
    assert torch.equal(o16.cpu(), torch.from_numpy(prod).to(torch.bfloat16))


@pytest.mark.parametrize("prec,tol", [("exact", EXACT_REL_FRO), ("fast", 3e-4)])
def test_per_group_heavy_tails(fq, prec, tol):
    """Per-group scales on the INT8 matrix cores (csrc/fql_group_i8.h) walk the residual limb set of flagged rows as the
    per-row kernels do: outlier channels stay inside the mode's constant, row by row."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(123)
    Kg, Ng, Bg, group = 1024, 200, 96, 128
    w = rng.standard_normal((Ng, Kg)).astype(np.float32)
    p, s, z = O.quantize_weights_grouped(w, group)
    x = rng.standard_normal((Bg, Kg)).astype(np.float32)
    x[::3, [7, 500]] *= 700.0                                  # every third row heavy-tailed, the others clean
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision=prec).cpu().numpy()
    ref = O.reference_linear_grouped(x, p, s, z)
    for t in range(Bg):
        assert rel_fro(out[t], ref[t]) < tol, (t, rel_fro(out[t], ref[t]))
