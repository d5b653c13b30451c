"""``QuantizedLinear`` -- nn.Module with the reference's exact surface (python/module.py:33-138):
constructor ``(in_features, out_features)``, buffers ``packed_weights [N, K//2] uint8``,
``scales [N] float32``, ``zero_points [N] float32`` (state_dict keys and shapes unchanged),
``from_linear``, ``forward`` for 1-D / 2-D float32 input, ``extra_repr``.

GPU tensors run the fused HIP kernels of libfql_int4.so (GEMV for B <= 2, INT8-limb MFMA GEMM
otherwise); if the extension is not built that raises -- there is no silent fallback.  CPU tensors
take the un-fused dequantize-then-matmul, exactly as the reference's ``forward`` does.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .quantize import quantize_weights, reference_quantized_linear


class QuantizedLinear(nn.Module):
    def __init__(self, in_features: int, out_features: int, precision: str = "default", bias: bool = False,
                 group_size: int | None = None):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.precision = precision
        # not in the reference (per-row only, python/quantize.py:73-80): per-group scales along K, buffers [N, K / group_size]
        if group_size is not None and (group_size <= 0 or group_size % 2 != 0 or in_features % group_size != 0):
            raise ValueError("group_size must be positive, even and divide in_features")
        self.group_size = None if group_size in (None, in_features) else group_size
        # not in the reference (it asserts `bias is None`, python/module.py:84): an optional float32 bias, added in the
        # kernels' epilogues.  Without one the module's state_dict is exactly the reference's three buffers.
        if bias:
            self.register_buffer("bias", torch.zeros(out_features, dtype=torch.float32))
        else:
            self.bias = None
        self.register_buffer("packed_weights", torch.zeros(out_features, in_features // 2, dtype=torch.uint8))
        sz_shape = (out_features,) if self.group_size is None else (out_features, in_features // self.group_size)
        self.register_buffer("scales", torch.zeros(sz_shape, dtype=torch.float32))
        self.register_buffer("zero_points", torch.zeros(sz_shape, dtype=torch.float32))

    @classmethod
    def from_linear(cls, linear: nn.Linear, precision: str = "default", group_size: int | None = None) -> "QuantizedLinear":
        """Quantise an ``nn.Linear``.  (The reference asserts there is no bias, python/module.py:84; here a bias is kept
        as a float32 buffer and added after the quantised matmul.)"""
        module = cls(linear.in_features, linear.out_features, precision=precision, bias=linear.bias is not None,
                     group_size=group_size)
        if linear.bias is not None:
            module.bias = linear.bias.data.detach().to(torch.float32).clone()
        packed, scales, zero_points = quantize_weights(linear.weight.data, group_size=module.group_size)
        module.packed_weights = packed
        module.scales = scales
        module.zero_points = zero_points
        return module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.is_cuda:
            return self._forward_cuda(x)
        out = reference_quantized_linear(x, self.packed_weights, self.scales, self.zero_points)
        return out if self.bias is None else out + self.bias

    def _forward_cuda(self, x: torch.Tensor) -> torch.Tensor:
        from . import ops
        return ops.linear_forward(x, self.packed_weights, self.scales, self.zero_points,
                                  precision=self.precision, bias=self.bias)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bits=4"
