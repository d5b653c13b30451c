"""fused INT4 dequantize-linear / MoE expert GEMM for AMD Instinct MI355X (gfx950).

Drop-in for the Python surface of samy19980109/Fused-4-bit-Dequantize-Linear-CUDA-Kernel
(reference python/__init__.py:14-22 exports, plus the MoE modules the reference keeps in
python/moe_int4_module.py and benchmark/moe_grouped_gemm/).
"""
from .quantize import quantize_weights, dequantize_weights, reference_quantized_linear
from .module import QuantizedLinear
from .moe import MoEINT4, quantize_weights_moe, QuantizedMoE, QuantizedMoEExpert, QuantizedMoEFFN
from .routing import (RoutingResult, simulate_routing, balanced_routing, create_expert_inputs,
                      combine_expert_outputs, dispatch_grouped, dispatch_indices, combine_grouped)

__all__ = [
    "quantize_weights", "dequantize_weights", "reference_quantized_linear", "QuantizedLinear",
    "MoEINT4", "quantize_weights_moe", "QuantizedMoE", "QuantizedMoEExpert", "QuantizedMoEFFN",
    "RoutingResult", "simulate_routing", "balanced_routing", "create_expert_inputs",
    "combine_expert_outputs", "dispatch_grouped", "dispatch_indices", "combine_grouped",
]
