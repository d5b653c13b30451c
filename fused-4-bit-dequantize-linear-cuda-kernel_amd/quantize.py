"""Weight format: asymmetric per-row INT4 with two weights per byte.

Same public functions, argument meaning and results (bit-exact, pinned by tests/golden) as the
reference's ``python/quantize.py``:

  quantize_weights            python/quantize.py:38-124
  dequantize_weights          python/quantize.py:127-173
  reference_quantized_linear  python/quantize.py:176-202

Format: ``packed[n, j] = (q[n, 2j+1] << 4) | q[n, 2j]`` (uint8, ``[N, K/2]``),
``w[n, k] = (q[n, k] - zero_points[n]) * scales[n]`` with float32 ``scales``/``zero_points`` of
shape ``[N]``.  These run on whatever device the tensors live on; on a GPU,
``dequantize_weights`` uses the HIP kernel of the extension.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

__all__ = ["quantize_weights", "dequantize_weights", "reference_quantized_linear",
           "pack_nibbles", "unpack_nibbles"]


def pack_nibbles(q: torch.Tensor) -> torch.Tensor:
    """[..., K] uint8 values in 0..15 -> [..., K/2] bytes, even index in the low nibble."""
    return (q[..., 1::2] << 4) | q[..., 0::2]


def unpack_nibbles(packed: torch.Tensor) -> torch.Tensor:
    """Inverse of :func:`pack_nibbles` (python/quantize.py:152-163)."""
    if packed.is_cuda:
        from . import ops
        return ops.unpack_nibbles(packed)
    q = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=torch.uint8, device=packed.device)
    q[..., 0::2] = packed & 0x0F
    q[..., 1::2] = packed >> 4
    return q


def quantize_weights(weight_fp32: torch.Tensor, num_bits: int = 4, group_size: int | None = None):
    """``[N, K]`` float32 -> ``(packed [N, K/2] uint8, scales [N], zero_points [N])``.

    Per row: ``scale = (max - min) / qmax``; a constant row uses ``max(|v|, 1) / qmax``; scale is
    floored at 1e-8; ``zp = clamp(round(-min / scale), 0, qmax)``;
    ``q = clamp(round(w / scale + zp), 0, qmax)`` with round-half-to-even.

    ``group_size`` (not in the reference): the same rule applied to every run of ``group_size``
    consecutive k of a row -- ``scales`` / ``zero_points`` become ``[N, K / group_size]`` (the GPTQ / AWQ layout).
    """
    assert weight_fp32.ndim == 2, "Weight must be 2D [output_dim, input_dim]"
    assert weight_fp32.shape[1] % 2 == 0, "input_dim must be even for packing"
    if group_size is not None and group_size != weight_fp32.shape[1]:
        N, K = weight_fp32.shape
        assert group_size > 0 and group_size % 2 == 0 and K % group_size == 0, "group_size must be even and divide input_dim"
        G = K // group_size
        p, s, z = quantize_weights(weight_fp32.reshape(N * G, group_size).contiguous(), num_bits)
        return p.reshape(N, K // 2), s.reshape(N, G), z.reshape(N, G)
    if weight_fp32.is_cuda and num_bits == 4 and weight_fp32.dtype == torch.float32:
        from . import ops                       # HIP quantiser: bit-exact with the host path below
        return ops.quantize_rows(weight_fp32)
    qmax = (1 << num_bits) - 1
    lo, hi = torch.aminmax(weight_fp32, dim=1)
    span_scale = (hi - lo) / qmax
    flat_scale = hi.abs().clamp(min=1.0) / qmax                 # rows whose values are all equal
    scales = torch.where(hi == lo, flat_scale, span_scale).clamp(min=1e-8)
    zero_points = torch.round(-lo / scales).clamp(0, qmax)
    q = torch.round(weight_fp32 / scales[:, None] + zero_points[:, None]).clamp(0, qmax).to(torch.uint8)
    return pack_nibbles(q), scales, zero_points


def dequantize_weights(packed_uint8: torch.Tensor, scales: torch.Tensor, zero_points: torch.Tensor):
    """``[N, K/2]`` packed bytes -> ``[N, K]`` float32: ``(q - zp) * scale``; ``scales`` / ``zero_points`` ``[N]``
    (per row, the reference's format) or ``[N, K / group_size]`` (per group along K)."""
    if scales.dim() == 2:                                   # per-group: the per-row rule on [N * G, group_size]
        N, G = scales.shape
        K = packed_uint8.shape[1] * 2
        w = dequantize_weights(packed_uint8.reshape(N * G, K // G // 2).contiguous(), scales.reshape(-1).contiguous(),
                               zero_points.reshape(-1).contiguous())
        return w.reshape(N, K)
    if packed_uint8.is_cuda:
        from . import ops
        return ops.dequantize_forward(packed_uint8, scales, zero_points)
    q = unpack_nibbles(packed_uint8).to(torch.float32)
    return (q - zero_points.unsqueeze(1)) * scales.unsqueeze(1)


def reference_quantized_linear(input: torch.Tensor, packed_weights: torch.Tensor,
                               scales: torch.Tensor, zero_points: torch.Tensor):
    """Un-fused formulation: materialise the float32 weights, then ``F.linear``.

    Part of the reference's public API (python/__init__.py:14-22).  It is what a CPU tensor gets
    from ``QuantizedLinear.forward`` (python/module.py:113-118); GPU tensors never come here --
    they go to the fused HIP kernels via :mod:`ops`.
    """
    return F.linear(input, dequantize_weights(packed_weights, scales, zero_points))
