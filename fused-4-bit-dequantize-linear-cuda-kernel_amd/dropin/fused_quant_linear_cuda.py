"""Import-name drop-in for the reference's pybind11 extension ``fused_quant_linear_cuda``
(csrc/quantized_linear.cpp:22-28).  Put this directory on ``sys.path`` and the reference's
unmodified ``python/module.py:129-132`` / ``tests/test_correctness.py:213-215`` call sites run
on the MI355X kernels:

    fused_quant_linear_cuda.forward(input, packed_weights, scales, zero_points) -> Tensor
"""
import fused_int4_amd as _pkg
from fused_int4_amd import ops as _ops


def forward(input, packed_weights, scales, zero_points):
    """Fused 4-bit dequantize + linear forward (HIP, gfx950)."""
    return _ops.linear_forward(input, packed_weights, scales, zero_points)
