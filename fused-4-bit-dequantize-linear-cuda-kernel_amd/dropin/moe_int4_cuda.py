"""Import-name drop-in for the reference's ``moe_int4_cuda`` extension
(csrc/moe_int4_kernel.cu:138-141):

    moe_int4_cuda.forward(packed_weights, scales, zero_points, inputs, expert_ids,
                          tokens_per_expert, input_offsets) -> Tensor
"""
import fused_int4_amd as _pkg
from fused_int4_amd import ops as _ops


def forward(packed_weights, scales, zero_points, inputs, expert_ids, tokens_per_expert, input_offsets):
    """Fused MoE INT4 forward (HIP, gfx950): one launch sequence for all experts."""
    return _ops.moe_forward(packed_weights, scales, zero_points, inputs, expert_ids,
                            tokens_per_expert, input_offsets)
