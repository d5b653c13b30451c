"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not product code.

A CPU restatement of the reference's algorithm for the fused INT4 dequantize-linear /
MoE expert-GEMM hot path.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this file; the shipped package
(``fused-4-bit-dequantize-linear-cuda-kernel_amd/``) never does.

Parity status: PINNED.  Every function below is checked in ``tests/test_oracle_golden.py``
against vectors produced by running the reference's own Python in the build container
(``tests/golden/*.npz``, generator ``tests/golden/make_golden.py``): packed bytes, scales
and zero-points bit-exact; dequantised weights bit-exact; linear outputs to 1e-6.

Each function cites the reference lines it restates (paths relative to the reference
repository root).  The arithmetic is numpy float32 / uint8 so that it runs anywhere
numpy does; the matmul of the oracle proper uses torch's ``F.linear`` when torch is
importable (that is what the reference calls) and ``numpy.matmul`` otherwise.
"""
from __future__ import annotations

import numpy as np

try:  # torch is optional for the oracle; the reference's matmul is F.linear
    import torch
    import torch.nn.functional as _F
except Exception:  # pragma: no cover
    torch = None

F32 = np.float32


# --------------------------------------------------------------------------------------
# A1  quantize_weights            reference: python/quantize.py:38-124
# --------------------------------------------------------------------------------------
def quantize_weights(weight_fp32, num_bits: int = 4):
    """Asymmetric per-ROW 4-bit quantisation + nibble packing.

    python/quantize.py:63-64   2-D, even K asserted
    python/quantize.py:73-80   row min/max, scale = (max-min)/15           (float32)
    python/quantize.py:85-94   constant rows: scale = max(|v|,1)/15; clamp(scale, 1e-8)
    python/quantize.py:100-101 zp = clamp(round(-min/scale), 0, 15)        (half-to-even)
    python/quantize.py:106-109 q  = clamp(round(w/scale + zp), 0, 15) -> uint8
    python/quantize.py:120-122 byte j = (q[2j+1] << 4) | q[2j]
    """
    w = np.ascontiguousarray(np.asarray(weight_fp32, dtype=F32))
    assert w.ndim == 2, "Weight must be 2D [output_dim, input_dim]"
    assert w.shape[1] % 2 == 0, "input_dim must be even for packing"
    max_val = F32((1 << num_bits) - 1)
    w_min = w.min(axis=1)
    w_max = w.max(axis=1)
    scales = (w_max - w_min) / max_val
    constant = w_max == w_min
    safe = np.where(constant, np.maximum(np.abs(w_max), F32(1.0)) / max_val, scales).astype(F32)
    safe = np.maximum(safe, F32(1e-8))
    zp = np.rint(-w_min / safe).astype(F32)
    zp = np.clip(zp, F32(0), max_val)
    q = np.rint(w / safe[:, None] + zp[:, None])
    q = np.clip(q, F32(0), max_val).astype(np.uint8)
    packed = ((q[:, 1::2] << 4) | q[:, 0::2]).astype(np.uint8)
    return packed, safe.astype(F32), zp.astype(F32)


# --------------------------------------------------------------------------------------
# A2  dequantize_weights          reference: python/quantize.py:127-173
# --------------------------------------------------------------------------------------
def unpack_nibbles(packed):
    """python/quantize.py:152-163: even index <- low nibble, odd index <- high nibble."""
    p = np.asarray(packed, dtype=np.uint8)
    out = np.empty(p.shape[:-1] + (p.shape[-1] * 2,), dtype=np.uint8)
    out[..., 0::2] = p & 0x0F
    out[..., 1::2] = p >> 4
    return out


def dequantize_weights(packed, scales, zero_points):
    """python/quantize.py:172: (w_int - zp[:,None]) * scale[:,None], all float32."""
    q = unpack_nibbles(packed).astype(F32)
    s = np.asarray(scales, dtype=F32)
    z = np.asarray(zero_points, dtype=F32)
    return ((q - z[..., None]) * s[..., None]).astype(F32)


# --------------------------------------------------------------------------------------
# A3  reference_quantized_linear  reference: python/quantize.py:176-202   (THE ORACLE)
# --------------------------------------------------------------------------------------
def reference_quantized_linear(x, packed, scales, zero_points, exact: bool = False):
    """dequantize_weights -> F.linear (python/quantize.py:201-202).

    ``exact=True`` accumulates the same float32 dequantised weights in float64 -- a
    tighter ground truth than either sgemm order, used when judging *which* of two
    float32 results is closer.
    """
    w = dequantize_weights(packed, scales, zero_points)
    x = np.asarray(x, dtype=F32)
    if exact:
        return (x.astype(np.float64) @ w.astype(np.float64).T)
    if torch is not None:
        return _F.linear(torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(w)).numpy()
    return (x @ w.T).astype(F32)


# --------------------------------------------------------------------------------------
# A8  quantize_weights_moe        reference: python/moe_int4_module.py:19-80
# --------------------------------------------------------------------------------------
def quantize_weights_moe(weights_list):
    """Per-TENSOR (per-expert) asymmetric quantisation, broadcast into [E,N] rows.

    python/moe_int4_module.py:45-47  w.float(); global min/max
    python/moe_int4_module.py:49     scale = ((max-min)/15).item()      (float32 value)
    python/moe_int4_module.py:50-51  zp = clamp(python round(-min/scale), 0, 15)
    python/moe_int4_module.py:57-59  q = clamp(round(w/scale + zp), 0, 15)
    python/moe_int4_module.py:62-76  same nibble order as A1
    No zero-range guard in the reference (constant tensor -> division by zero); kept.
    """
    E = len(weights_list)
    N, K = weights_list[0].shape
    packed = np.zeros((E, N, K // 2), dtype=np.uint8)
    scales = np.zeros((E, N), dtype=F32)
    zps = np.zeros((E, N), dtype=F32)
    for e, w in enumerate(weights_list):
        w32 = np.asarray(w).astype(F32)
        w_min = w32.min()
        w_max = w32.max()
        scale = F32((w_max - w_min) / F32(15.0))
        zp = float(np.rint(np.float64(F32(-w_min) / scale)))      # python round == half-to-even
        zp = max(0.0, min(15.0, zp))
        scales[e] = scale
        zps[e] = zp
        q = np.clip(np.rint(w32 / scale + F32(zp)), 0, 15).astype(np.uint8)
        packed[e] = (q[:, 1::2] << 4) | q[:, 0::2]
    return packed, scales, zps


# --------------------------------------------------------------------------------------
# A6 (intended semantics) / A9   grouped per-expert GEMM
#   intended: csrc/moe_int4_kernel.cu:93-136 (host loop), python/moe_int4_module.py:122-146
#   equals QuantizedMoE.forward: benchmark/moe_grouped_gemm/moe_int4_module.py:63-72,123-125
# --------------------------------------------------------------------------------------
def reference_moe_grouped(inputs, packed, scales, zero_points, tokens_per_expert, input_offsets,
                          exact: bool = False):
    """out[off_e : off_e+cnt_e] = inputs[off_e : off_e+cnt_e] @ dequant(W_e)^T ; other rows 0.

    The reference's CUDA kernel is defective; the contract is the one its host
    wrapper and Python module document: rows pre-grouped by expert, ``expert_ids`` unused
    (csrc/moe_int4_kernel.cu:98), output zero-initialised (``torch::zeros`` :109).
    """
    x = np.asarray(inputs, dtype=F32)
    T = x.shape[0]
    E, N, _ = np.asarray(packed).shape
    out = np.zeros((T, N), dtype=np.float64 if exact else F32)
    for e in range(E):
        cnt = int(tokens_per_expert[e])
        off = int(input_offsets[e])
        if cnt <= 0:
            continue
        out[off:off + cnt] = reference_quantized_linear(
            x[off:off + cnt], packed[e], scales[e], zero_points[e], exact=exact)
    return out


def quantized_moe_forward(expert_inputs, packed_list, scales_list, zp_list):
    """QuantizedMoE.forward(List[Tensor]) -> List[Tensor]
    benchmark/moe_grouped_gemm/moe_int4_module.py:63-72: empty x -> empty float16 [0,N];
    else x @ dequant(W).T cast to x.dtype (float16 inputs are multiplied in float16 by torch;
    here the product is formed in float32 and rounded once to x.dtype)."""
    outs = []
    for x, p, s, z in zip(expert_inputs, packed_list, scales_list, zp_list):
        x = np.asarray(x)
        N = np.asarray(p).shape[0]
        if x.shape[0] == 0:
            outs.append(np.empty((0, N), dtype=np.float16))
            continue
        y = reference_quantized_linear(x.astype(F32), p, s, z)
        outs.append(np.asarray(y).astype(x.dtype))
    return outs


# --------------------------------------------------------------------------------------
# A10  dispatch / combine          reference: benchmark/moe_grouped_gemm/routing.py:96-189
# --------------------------------------------------------------------------------------
def create_expert_inputs(x, expert_indices, num_experts):
    """routing.py:117-149: flatten [T,top_k], sort by expert, gather rows, inverse permutation.
    The reference's argsort is not stable; only the *set* of rows per expert and the
    round-trip through ``combine`` are defined, so a stable sort is used here."""
    x = np.asarray(x)
    idx = np.asarray(expert_indices)
    T, top_k = idx.shape
    flat_token = np.repeat(np.arange(T), top_k)
    flat_expert = idx.reshape(-1)
    order = np.argsort(flat_expert, kind="stable")
    inverse = np.argsort(order, kind="stable")
    counts = np.bincount(flat_expert, minlength=num_experts)
    sorted_tokens = flat_token[order]
    grouped = x[sorted_tokens]
    offsets = np.concatenate([[0], np.cumsum(counts)[:-1]])
    return grouped, counts.astype(np.int32), offsets.astype(np.int32), inverse


def combine_expert_outputs(grouped_out, expert_weights, inverse, top_k):
    """routing.py:172-189: unsort -> [T, top_k, N] -> weighted sum over top_k."""
    y = np.asarray(grouped_out)[inverse]
    T = y.shape[0] // top_k
    y = y.reshape(T, top_k, -1)
    return (y * np.asarray(expert_weights)[..., None]).sum(axis=1)


# --------------------------------------------------------------------------------------
# A11  the reference's own roofline byte/flop model   benchmark/run_benchmark.py:222,227
# --------------------------------------------------------------------------------------
def reference_roofline_model(K: int, N: int):
    bytes_read = K * 4 + N * (K // 2) + N * 4 + N * 4
    flops = 2 * K * N
    return bytes_read, flops


def gated_ffn_grouped(gate_up_q, down_q, x, counts, offsets):
    """CPU restatement of a gated FFN expert over rows grouped by expert (no reference
    counterpart: the reference models the up projection only, benchmark/moe_grouped_gemm/config.py:50-52):
    ``out = Wd_e @ (silu(Wg_e @ x) * (Wu_e @ x))`` with the de-quantised weights, float64 accumulation.
    ``gate_up_q`` / ``down_q``: (packed [E, rows, cols/2], scales [E, rows], zero_points [E, rows])."""
    Pg, Sg, Zg = gate_up_q
    Pd, Sd, Zd = down_q
    E, F2, _ = Pg.shape
    F = F2 // 2
    H = Pd.shape[1]
    out = np.zeros((x.shape[0], H), dtype=np.float64)
    for e in range(E):
        c, o = int(counts[e]), int(offsets[e])
        if c == 0:
            continue
        wgu = dequantize_weights(Pg[e], Sg[e], Zg[e]).astype(np.float64)
        wd = dequantize_weights(Pd[e], Sd[e], Zd[e]).astype(np.float64)
        gu = x[o:o + c].astype(np.float64) @ wgu.T
        g, u = gu[:, :F], gu[:, F:]
        h = (g / (1.0 + np.exp(-g))) * u
        out[o:o + c] = h @ wd.T
    return out


# --------------------------------------------------------------------------------------
# fp8 (OCP e4m3fn) activations -- BASELINE.json configs[4].  NOT in the reference: it lists FP8 as
# future work only (README.md:228), so there is no reference vector to pin this to ("parity unpinned"
# by the reference).  The FORMAT is pinned instead: tests/test_oracle_golden.py checks the table and
# the rounding below against torch's own float8_e4m3fn casts, and the arithmetic is the same
# dequantize-then-matmul (python/quantize.py:176-202) applied to the decoded activations in float64.
# --------------------------------------------------------------------------------------
def e4m3_table():
    """All 256 OCP e4m3fn values: 1 sign, 4 exponent (bias 7), 3 mantissa bits; no infinities;
    S.1111.111 is NaN; exponent field 0 is subnormal (m * 2^-9)."""
    b = np.arange(256, dtype=np.int64)
    e = (b >> 3) & 15
    m = b & 7
    mag = np.where(e == 0, m * 2.0 ** -9, (1.0 + m / 8.0) * 2.0 ** (e - 7.0))
    mag = np.where((e == 15) & (m == 7), np.nan, mag)
    return np.where(b >= 128, -mag, mag).astype(F32)


def e4m3_decode(xbytes):
    return e4m3_table()[np.asarray(xbytes, dtype=np.uint8)]


def e4m3_encode(x):
    """float32 -> e4m3fn byte, round to nearest, ties to even mantissa; magnitudes that round above 448 (> 464)
    and NaN become NaN (0x7F | sign), as torch's cast does."""
    x = np.asarray(x, dtype=F32)
    tab = e4m3_table()[:127].astype(np.float64)            # the 127 finite non-negative values, ascending (0x00..0x7E)
    a = np.abs(x).astype(np.float64)
    hi = np.clip(np.searchsorted(tab, a, side="left"), 1, 126)
    lo = hi - 1
    dlo, dhi = a - tab[lo], tab[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((hi & 1) == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    code = np.where(a > 464.0, 0x7F, code)                  # past the midpoint between 448 and the absent 480 (the tie goes to 448)
    code = np.where(np.isnan(x), 0x7F, code)
    return (code | (np.signbit(x).astype(np.uint8) << 7)).astype(np.uint8)


def quantize_activations_fp8(x):
    """Per-row e4m3 quantisation of the library's FQL_PRECISION_FP8 mode: scale[t] = max|x[t]| / 448 in float32
    (1 for an all-zero row), byte = e4m3(x / scale) with a float32 division."""
    x = np.ascontiguousarray(np.asarray(x, dtype=F32))
    amax = np.abs(x).max(axis=1)
    scale = np.where(amax == 0, F32(1.0), amax / F32(448.0)).astype(F32)
    return e4m3_encode((x / scale[:, None]).astype(F32)), scale


def reference_linear_fp8(xbytes, act_scale, packed, scales, zero_points):
    """out = (act_scale * e4m3(x)) @ dequantize_weights(...).T, accumulated in float64."""
    a = e4m3_decode(xbytes).astype(np.float64)
    if act_scale is not None:
        a = a * np.asarray(act_scale, dtype=np.float64)[:, None]
    w = dequantize_weights(packed, scales, zero_points).astype(np.float64)
    return a @ w.T


def reference_moe_grouped_fp8(xbytes, act_scale, packed, scales, zero_points, tokens_per_expert, input_offsets):
    T = np.asarray(xbytes).shape[0]
    N = np.asarray(packed).shape[1]
    out = np.zeros((T, N), dtype=np.float64)
    for e, (c, o) in enumerate(zip(np.asarray(tokens_per_expert), np.asarray(input_offsets))):
        lo, hi = max(int(o), 0), min(int(o) + int(c), T)
        if hi > lo:
            sc = None if act_scale is None else np.asarray(act_scale)[lo:hi]
            out[lo:hi] = reference_linear_fp8(np.asarray(xbytes)[lo:hi], sc, packed[e], scales[e], zero_points[e])
    return out


# --------------------------------------------------------------------------------------
# per-GROUP scales along K.  NOT in the reference (per row only, python/quantize.py:73-80): the
# reference's per-row rule (quantize_weights above, pinned) applied to every run of `group_size` consecutive k.
# --------------------------------------------------------------------------------------
def quantize_weights_grouped(weight_fp32, group_size):
    w = np.ascontiguousarray(np.asarray(weight_fp32, dtype=F32))
    N, K = w.shape
    assert group_size % 2 == 0 and K % group_size == 0
    G = K // group_size
    p, s, z = quantize_weights(w.reshape(N * G, group_size))
    return p.reshape(N, K // 2), s.reshape(N, G), z.reshape(N, G)


def dequantize_weights_grouped(packed, scales, zero_points):
    packed = np.asarray(packed, dtype=np.uint8)
    N, G = np.asarray(scales).shape
    K = packed.shape[1] * 2
    q = unpack_nibbles(packed).astype(F32).reshape(N, G, K // G)
    return ((q - np.asarray(zero_points, F32)[:, :, None]) * np.asarray(scales, F32)[:, :, None]).reshape(N, K)


def reference_linear_grouped(x, packed, scales, zero_points):
    """dequantize-then-matmul (python/quantize.py:176-202) with per-group constants, accumulated in float64."""
    return np.asarray(x, np.float64) @ dequantize_weights_grouped(packed, scales, zero_points).astype(np.float64).T
