"""GPU tests of the one-wave-per-SIMD GEMM (csrc/fql_gemm_w4.h; tuning ids 300..): bit for bit the results of the
8-wave wide kernel (tuning id 0) on the same limbs, and within the exact mode's constant of the float64 oracle.

Replaces (reference): csrc/moe_int4_kernel.cu:17-136, csrc/quantized_linear_kernel.cu:90-279."""
import numpy as np
import pytest
import torch

from helpers import EXACT_REL_FRO, rel_fro
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


def w4_ids():
    from fused_int4_amd import _native
    return list(range(300, 300 + _native.lib().fql_tune_num_w4_configs()))


def make_moe(E, N, K, counts, seed, heavy_every=0):
    rng = np.random.default_rng(seed)
    q = [O.quantize_weights((rng.standard_normal((N, K)) * 0.02).astype(np.float32)) for _ in range(E)]
    P, S, Z = (np.stack([t[i] for t in q]) for i in range(3))
    counts = np.asarray(counts, np.int32)
    offs = (np.cumsum(counts) - counts).astype(np.int32)
    T = int(counts.sum())
    x = rng.standard_normal((T, K)).astype(np.float32)
    if heavy_every:
        for t in range(0, T, heavy_every):
            x[t, rng.choice(K, 2, replace=False)] *= 800.0
    return P, S, Z, x, counts, offs


def run_cfgs(ops, cfgs, limbs, delta, rowsum, dP, dS, dZ, dc, do, E, T, K, N):
    outs = {}
    for cfg in cfgs:
        out = torch.full((T, N), float("nan"), dtype=torch.float32, device="cuda")
        rc = ops.tune_gemm_i8(cfg, limbs, delta, rowsum, dP, dS, dZ, dc, do, out, E, T, K, N, "exact")
        assert rc == 0, (cfg, rc)
        torch.cuda.synchronize()
        outs[cfg] = out
    return outs


@pytest.mark.parametrize("E,N,K,counts,heavy", [
    (8, 352, 512, [128] * 8, 0),                             # 11 fragments: one 6- and one 5-fragment tile; two weight stages (the minimum)
    (5, 200, 768, [0, 7, 33, 70, 129], 0),                   # ragged groups, N % 32 != 0, 4- and 3-fragment tiles
    (6, 1100, 1024, [128, 1, 96, 64, 130, 32], 0),           # row blocks without rows: waves computing on zeros
    (5, 200, 768, [0, 7, 33, 70, 129], 5),                   # heavy-tailed rows: residual pass + main pass
    (4, 2080, 1280, [200, 128, 56, 128], 7),                 # several tiles per workgroup with residual visits in between
    (3, 384, 4128, [100, 128, 12], 0),                       # K % 256 != 0 (zero-padded last stage), 17 stages
])
def test_w4_bit_identical_to_wide_kernel(fq, E, N, K, counts, heavy):
    from fused_int4_amd import ops
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, 1234 + N, heavy)
    T = x.shape[0]
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision="exact", tokens_per_expert=dc, input_offsets=do)
    if heavy:
        assert 0 < int((delta[1] != 0).sum()) < T
    outs = run_cfgs(ops, [0] + w4_ids(), limbs, delta, rowsum, dP, dS, dZ, dc, do, E, T, K, N)
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert rel_fro(outs[0].cpu().numpy(), ref) < EXACT_REL_FRO
    for cfg in w4_ids():
        assert torch.equal(outs[cfg], outs[0]), f"configuration {cfg} differs from the wide kernel"


def test_w4_linear_many_tiles_per_workgroup(fq):
    """One matrix, 8192 rows x 12480 columns: 64 x 65 = 4160 tiles on at most 256 workgroups, so every workgroup walks
    more than the 16 tiles its LDS table describes at a time (the refill path)."""
    from fused_int4_amd import ops
    rng = np.random.default_rng(5)
    N, K, T = 12480, 512, 8192
    p, s, z = O.quantize_weights((rng.standard_normal((N, K)) * 0.02).astype(np.float32))
    x = rng.standard_normal((T, K)).astype(np.float32)
    dP, dS, dZ, dx = dev(p), dev(s), dev(z), dev(x)
    limbs, delta, rowsum = ops.act_quant(dx, precision="exact")
    outs = run_cfgs(ops, [0] + w4_ids(), limbs, delta, rowsum, dP, dS, dZ, None, None, 1, T, K, N)
    rows = [0, 127, 128, 4097, 8191]
    ref = C.linear_f64acc(x[rows], p, s, z)
    assert rel_fro(outs[0][rows].cpu().numpy(), ref) < EXACT_REL_FRO
    for cfg in w4_ids():
        assert torch.equal(outs[cfg], outs[0]), f"configuration {cfg} differs from the wide kernel"


def test_w4_headline_shape_bit_identical(fq):
    """BASELINE configs[2] at full size (8 experts, 4096 -> 11008, 1024 routed rows), balanced and skewed routing."""
    from fused_int4_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    E, K, N = 8, 4096, 11008
    q = [fq.quantize_weights(torch.randn(N, K, device="cuda", generator=g) * 0.02) for _ in range(E)]
    dP, dS, dZ = (torch.stack([t[i] for t in q]) for i in range(3))
    for counts in ([128] * 8, [485, 230, 140, 90, 45, 24, 6, 4]):
        cnt = torch.tensor(counts, dtype=torch.int32, device="cuda")
        offs = (torch.cumsum(cnt, 0) - cnt).to(torch.int32)
        T = int(sum(counts))
        x = torch.randn(T, K, device="cuda", generator=g)
        limbs, delta, rowsum = ops.act_quant(x, precision="exact", tokens_per_expert=cnt, input_offsets=offs)
        outs = run_cfgs(ops, [0] + w4_ids(), limbs, delta, rowsum, dP, dS, dZ, cnt, offs, E, T, K, N)
        for cfg in w4_ids():
            assert torch.equal(outs[cfg], outs[0]), f"configuration {cfg} differs from the wide kernel ({counts})"


@pytest.mark.parametrize("counts", [
    [485, 230, 140, 90, 45, 24, 6, 4],                       # the reference's default "skewed" routing at 512 tokens, top-2
    [128, 129, 192, 193, 64, 65, 1, 0],                      # remainders 0, 1, 64 (split off) and 65 (kept)
    [300, 0, 0, 17, 256, 63, 700, 5],
])
def test_short_row_groups_equal_wide_kernel(fq, counts):
    """Skewed routing: row tiles of at most 64 / 32 rows run with 3 / 2 fragments per wave (csrc/fql_gemm_w4.h, tile classes)
    and the tiles are walked in cost order.  Every row must carry the bits of the 8-wave wide kernel, through the product
    entry points too, and the small groups must match the float64 oracle."""
    from fused_int4_amd import ops
    E, N, K = len(counts), 1100, 1024
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, 99 + sum(counts), heavy_every=11)
    T = x.shape[0]
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision="exact", tokens_per_expert=dc, input_offsets=do)
    ref_out = run_cfgs(ops, [0], limbs, delta, rowsum, dP, dS, dZ, dc, do, E, T, K, N)[0]
    prod = ops.moe_forward(dP, dS, dZ, dx, None, dc, do)
    assert torch.equal(prod, ref_out)
    two_phase = ops.gemm_i8(limbs, delta, rowsum, dP, dS, dZ, dc, do, precision="exact")
    assert torch.equal(two_phase, ref_out)
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    got = prod.cpu().numpy()
    for e, (o, c) in enumerate(zip(offs, cnt)):
        if 0 < c <= 64:
            assert rel_fro(got[o:o + c], ref[o:o + c]) < EXACT_REL_FRO, e


def test_quantized_moe_keeps_one_copy_of_the_weights(fq):
    """QuantizedMoE (benchmark/moe_grouped_gemm/moe_int4_module.py:84-130): the grouped launch needs [E, N, K/2]; the
    experts' registered buffers become views of that one stacked storage instead of a second copy, state_dict keys and
    in-place loads keep working, and total_memory_bytes is what the module holds."""
    torch.manual_seed(0)
    E, K, N = 4, 256, 384
    ws = [torch.randn(N, K) * 0.02 for _ in range(E)]
    moe = fq.QuantizedMoE.from_fp16_weights([w.half() for w in ws]).cuda()
    keys = sorted(moe.state_dict().keys())
    xs = [torch.randn(m, K, device="cuda") for m in (5, 0, 70, 33)]
    out1 = moe(xs)
    st = moe._stacked[0]
    for i, e in enumerate(moe.experts):
        assert e.packed_weights.data_ptr() == st[i].data_ptr()               # a view of the stacked storage
        assert e.packed_weights.untyped_storage().data_ptr() == st.untyped_storage().data_ptr()
    assert sorted(moe.state_dict().keys()) == keys
    held = sum(b.untyped_storage().nbytes() for b in {id(b.untyped_storage()): b for b in moe.buffers()}.values())
    assert held == moe.total_memory_bytes
    # in-place load of other weights goes through the views: the next forward uses them, without re-stacking
    other = fq.QuantizedMoE.from_fp16_weights([(w * 1.5).half() for w in ws]).cuda()
    moe.load_state_dict(other.state_dict())
    assert moe._stacked[0].data_ptr() == st.data_ptr()
    out2 = moe(xs)
    ref2 = other(xs)
    for a, b in zip(out2, ref2):
        assert torch.equal(a, b)
    assert not torch.equal(out1[2], out2[2])


@pytest.mark.parametrize("tokens,E,K,N,top_k", [(512, 8, 1024, 1100, 2), (40, 4, 256, 200, 2), (33, 6, 512, 96, 1)])
def test_routing_weight_in_the_gemm_epilogue_equals_weighted_combine(fq, tokens, E, K, N, top_k):
    """SURVEY 8f N1, second half (reference: benchmark/moe_grouped_gemm/routing.py:172-189 multiplies by the routing weight
    after the expert GEMMs): the weight folded into the GEMM epilogue (fql_moe_gather_scaled_fwd_f32) + a pure gather-add
    must give the bits of fql_combine_f32 with the weights, for top_k <= 2; every tile class of the big case is hit."""
    from fused_int4_amd import ops, routing as R
    from fused_int4_amd.ep import ExpertParallelMoE
    g = torch.Generator(device="cuda").manual_seed(tokens + N)
    q = [fq.quantize_weights(torch.randn(N, K, device="cuda", generator=g) * 0.02) for _ in range(E)]
    P, S, Z = (torch.stack([t[i] for t in q]) for i in range(3))
    x = torch.randn(tokens, K, device="cuda", generator=g)
    route = R.simulate_routing(tokens, E, top_k, "skewed", torch.device("cuda"), 7)
    ep = ExpertParallelMoE(E, P, S, Z)
    ref = ep(x, route.expert_indices, route.expert_weights)
    ep.fold_weights = True
    got = ep(x, route.expert_indices, route.expert_weights)
    assert torch.equal(got, ref)
    # and the scaled rows themselves are (un-scaled row) * weight, one float32 rounding
    counts, offsets, token_of_sorted, pos_of_slot = ops.route_plan(route.expert_indices, E)
    w_sorted = torch.empty(pos_of_slot.numel(), device="cuda")
    w_sorted[pos_of_slot.long()] = route.expert_weights.reshape(-1).float()
    y = ops.moe_gather_forward(P, S, Z, x, token_of_sorted, counts, offsets)
    yw = ops.moe_gather_forward(P, S, Z, x, token_of_sorted, counts, offsets, row_weight=w_sorted)
    assert torch.equal(yw, y * w_sorted[:, None])


@pytest.mark.parametrize("counts,heavy", [
    ([200, 3, 40, 0, 129, 64, 33, 7, 1, 130], 0),           # every class after every other one in a workgroup's walk
    ([60, 150, 20, 140, 10, 50], 7),                        # the same with heavy-tailed rows (residual passes in between)
])
def test_visit_sequences_of_mixed_tile_classes(fq, counts, heavy):
    """A workgroup that walks MANY tiles of mixed row counts: the prefetch rings run across the visit boundaries
    (csrc/fql_gemm_w4.h: the short tile classes keep a full stage of activation fragments in flight, the 128-row class D
    k-steps, and hand over in a fixed state), so every order of classes must leave every row with the wide kernel's bits.
    The persistent grid is shrunk to 8 workgroups (tuning hook) to get those walks on a small shape."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    E, N, K = len(counts), 1500, 1536
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, 7 + sum(counts), heavy_every=heavy)
    T = x.shape[0]
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    limbs, delta, rowsum = ops.act_quant(dx, precision="exact", tokens_per_expert=dc, input_offsets=do)
    ref_out = run_cfgs(ops, [0], limbs, delta, rowsum, dP, dS, dZ, dc, do, E, T, K, N)[0]
    for cus in (8, 16, 24):
        old = lib.fql_tune_set_compute_units(cus)
        try:
            outs = run_cfgs(ops, w4_ids(), limbs, delta, rowsum, dP, dS, dZ, dc, do, E, T, K, N)
        finally:
            lib.fql_tune_set_compute_units(old)
        for cfg, out in outs.items():
            bad = (~(out == ref_out).all(dim=1)).nonzero().flatten().tolist()
            assert not bad, f"configuration {cfg} at {cus} workgroups: rows {bad[:8]} differ from the wide kernel"
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert rel_fro(ref_out.cpu().numpy(), ref) < EXACT_REL_FRO


@pytest.mark.parametrize("counts,heavy,spin,cus", [
    ([128] * 8, 0, -1, 0),                                  # the headline's routing (at a smaller N / K)
    ([200, 3, 40, 0, 129, 64, 33, 7, 1, 130], 5, -1, 0),    # every tile class, heavy-tailed rows, an empty expert
    ([200, 3, 40, 0, 129, 64, 33, 7, 1, 130], 5, 0, 0),     # no polling at all: every workgroup quantises its tiles' rows itself
    ([60, 150, 20, 140, 10, 50], 7, -1, 64),                # 64 workgroups: several tiles and several row groups per workgroup
    ([60, 150, 20, 140, 10, 50], 7, 3, 64),                 # ... with a short poll budget (a mix of waiting and self-service)
])
def test_one_launch_form_is_bit_identical(fq, counts, heavy, spin, cus):
    """csrc/fql_gemm_w4.h, FUSED: the pre-pass as the GEMM kernel's first phase (row groups published through flags, each
    workgroup waits for the rows of its own tiles only and quantises them itself after a bounded number of polls).  Same
    limbs, same arithmetic: the product call must return the bits of the two-launch form, whatever the polling does."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    E, N, K = len(counts), 1500, 1536
    P, S, Z, x, cnt, offs = make_moe(E, N, K, counts, 11 + sum(counts), heavy_every=heavy)
    dP, dS, dZ, dx, dc, do = dev(P), dev(S), dev(Z), dev(x), dev(cnt), dev(offs)
    ref_out = ops.moe_forward(dP, dS, dZ, dx, None, dc, do)
    torch.cuda.synchronize()
    old_cus = lib.fql_tune_set_compute_units(cus) if cus else None
    old_spin = lib.fql_tune_set_fused_spin(spin) if spin >= 0 else None
    old = lib.fql_tune_set_fused(1)
    try:
        outs = [ops.moe_forward(dP, dS, dZ, dx, None, dc, do) for _ in range(3)]     # (fresh tokens: flags of the launch before must not count)
        torch.cuda.synchronize()
    finally:
        lib.fql_tune_set_fused(old)
        if old_spin is not None:
            lib.fql_tune_set_fused_spin(old_spin)
        if old_cus is not None:
            lib.fql_tune_set_compute_units(old_cus)
    if cus:                                                   # the reference at the same pretended device (tile widths follow the grid)
        lib.fql_tune_set_compute_units(cus)
        try:
            ref_out = ops.moe_forward(dP, dS, dZ, dx, None, dc, do)
            torch.cuda.synchronize()
        finally:
            lib.fql_tune_set_compute_units(old_cus)
    for out in outs:
        bad = (~(out == ref_out).all(dim=1)).nonzero().flatten().tolist()
        assert not bad, f"rows {bad[:8]} of the one-launch form differ from the two-launch form"
    ref = C.moe_grouped(P, S, Z, x, cnt, offs)
    assert rel_fro(ref_out.cpu().numpy(), ref) < EXACT_REL_FRO


def test_one_launch_form_linear_and_gather(fq):
    """The same through the linear entry point (no expert table) and through the fused-dispatch entry point (rows gathered
    by index, routing weight in the epilogue)."""
    from fused_int4_amd import ops, _native
    lib = _native.lib()
    rng = np.random.default_rng(3)
    N, K, B = 1100, 1024, 300
    p, s, z = O.quantize_weights((rng.standard_normal((N, K)) * 0.02).astype(np.float32))
    x = rng.standard_normal((B, K)).astype(np.float32)
    x[::13, 5] *= 900.0
    dp, ds, dz, dx = dev(p), dev(s), dev(z), dev(x)
    ref_lin = ops.linear_forward(dx, dp, ds, dz)
    E = 4
    P, S, Z, xt, cnt, offs = make_moe(E, N, K, [70, 10, 130, 45], 5)
    tokens = dev(rng.standard_normal((100, K)).astype(np.float32))
    T = int(cnt.sum())
    idx = dev(rng.integers(0, 100, size=T).astype(np.int32))
    rw = dev(rng.random(T).astype(np.float32))
    dP, dS, dZ, dc, do = dev(P), dev(S), dev(Z), dev(cnt), dev(offs)
    ref_g = ops.moe_gather_forward(dP, dS, dZ, tokens, idx, dc, do, row_weight=rw)
    torch.cuda.synchronize()
    old = lib.fql_tune_set_fused(1)
    try:
        got_lin = ops.linear_forward(dx, dp, ds, dz)
        got_g = ops.moe_gather_forward(dP, dS, dZ, tokens, idx, dc, do, row_weight=rw)
        torch.cuda.synchronize()
    finally:
        lib.fql_tune_set_fused(old)
    assert torch.equal(got_lin, ref_lin)
    assert torch.equal(got_g, ref_g)
