"""Activation dynamic range: outlier channels and outlier rows (what LLM activations look like) through the
fixed-point MFMA path, against the float64-accumulating oracle.

The limbs are PER-ROW block fixed point (csrc/fql_act_quant.h): one power-of-two quantum delta[t] per row, set by
the row's largest magnitude, so one outlier element coarsens the quantum of its whole row.  Every element is
rounded once, by at most delta/2; for errors uncorrelated with the weights the output error of row t is

    ||out_t - ref_t|| / ||ref_t||  ~=  delta[t] / sqrt(12) * sqrt(K) / ||x_t||_2        (tests/helpers.row_quantum_bound)

which is <= 2^-(8L-2) * sqrt(K / 12) whatever the data (max|x_t| <= ||x_t||_2): 4.4e-6 at K = 4096 for 3 limbs.
The tests check (1) that bound, row by row, (2) the fixed constants of tests/helpers.py and the reference's own
allclose(atol) where the build promises them.  Reference tolerances: tests/test_correctness.py:218,233,252."""
import numpy as np
import pytest
import torch

from helpers import EXACT_REL_FRO, FAST_REL_FRO, rel_fro, row_quantum_bound
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
    bound = row_quantum_bound(x, 3)
    for t in range(B):
        assert rel_fro(out[t], ref[t]) < 2.0 * bound[t] + 3e-7, (t, rel_fro(out[t], ref[t]), bound[t])
    # the stated constant and the reference's tolerance hold for every case of this sweep
    assert rel_fro(out, ref) < EXACT_REL_FRO, rel_fro(out, ref)
    assert np.allclose(out, ref, atol=1e-2, rtol=1e-5), np.abs(out - ref).max()
    # not worse than a float32 FMA chain (the reference kernel's own arithmetic, csrc/quantized_linear_kernel.cu:240-244)
    f32 = C.linear_fma(x, p, s, z)
    assert rel_fro(out, ref) < 4.0 * rel_fro(f32, ref) + 3e-7


@pytest.mark.parametrize("kind,factor", CASES)
def test_fast_mode_heavy_tails(fq, weights, kind, factor):
    """2 limbs (15-bit fixed point per row): inside the north-star 1e-3 bound on every case of the sweep."""
    from fused_int4_amd import ops
    p, s, z = weights
    x = heavy_tailed(kind, factor, np.random.default_rng(int(factor) + len(kind)))
    out = ops.linear_forward(dev(x), dev(p), dev(s), dev(z), precision="fast").cpu().numpy()
    ref = C.linear_f64acc(x, p, s, z)
    bound = row_quantum_bound(x, 2)
    for t in range(B):
        assert rel_fro(out[t], ref[t]) < 2.0 * bound[t] + 3e-7, (t, rel_fro(out[t], ref[t]), bound[t])
    assert rel_fro(out, ref) < 1e-3, rel_fro(out, ref)


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
