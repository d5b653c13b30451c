"""MoE modules over the grouped INT4 GEMM.

``MoEINT4`` / ``quantize_weights_moe``  -- surface of the reference's python/moe_int4_module.py:19-146
    (stacked ``[E, N, K/2]`` buffers, per-TENSOR scale / zero-point broadcast to ``[E, N]``,
    ``forward(inputs, expert_ids, tokens_per_expert, input_offsets)`` over rows pre-grouped by expert).
``QuantizedMoEExpert`` / ``QuantizedMoE`` -- surface of benchmark/moe_grouped_gemm/moe_int4_module.py:21-130
    (per-expert per-ROW quantisation, ``forward(List[Tensor]) -> List[Tensor]``).

Both run ONE fused launch sequence for all experts on the GPU (device-side offsets, no per-expert
launches or host syncs).  The reference's ``moe_int4_cuda`` kernel is defective; the
semantics implemented are the ones its wrapper documents and ``QuantizedMoE.forward`` computes.
"""
from __future__ import annotations

from typing import List

import torch
import torch.nn as nn

from .quantize import quantize_weights, dequantize_weights, pack_nibbles

try:  # mirrors the reference's module-level availability flag (python/moe_int4_module.py:10-16)
    from . import _native
    _native.lib()
    CUDA_AVAILABLE = True
except Exception:  # library not built: MoEINT4.forward raises, as the reference does
    CUDA_AVAILABLE = False


def quantize_weights_moe(weights_list):
    """List of ``[N, K]`` (fp16/fp32) -> stacked ``(packed [E,N,K/2], scales [E,N], zero_points [E,N])``.

    Per expert, over the WHOLE tensor: ``scale = (max - min) / 15``, ``zp = clamp(round(-min/scale), 0, 15)``,
    ``q = clamp(round(w/scale + zp), 0, 15)`` (python/moe_int4_module.py:45-59).  As in the reference
    there is no guard for a constant tensor.
    """
    E = len(weights_list)
    N, K = weights_list[0].shape
    device = weights_list[0].device
    packed = torch.zeros(E, N, K // 2, dtype=torch.uint8, device=device)
    scales = torch.zeros(E, N, dtype=torch.float32, device=device)
    zero_points = torch.zeros(E, N, dtype=torch.float32, device=device)
    for e, w in enumerate(weights_list):
        w32 = w.float()
        if w32.is_cuda:                             # HIP quantiser: bit-exact with the host arithmetic below
            from . import ops
            packed[e], scales[e], zero_points[e] = ops.quantize_tensor(w32)
            continue
        lo, hi = torch.aminmax(w32)
        scale = ((hi - lo) / 15.0).item()
        zp = float(max(0, min(15, round((-lo / scale).item()))))
        scales[e] = scale
        zero_points[e] = zp
        q = torch.round(w32 / scale + zp).clamp(0, 15).to(torch.uint8)
        packed[e] = pack_nibbles(q)                 # one vectorised pack instead of a K/2-step loop
    return packed, scales, zero_points


class MoEINT4(nn.Module):
    def __init__(self, num_experts, hidden_dim, ffn_dim, precision: str = "default"):
        super().__init__()
        self.num_experts = num_experts
        self.hidden_dim = hidden_dim
        self.ffn_dim = ffn_dim
        self.packed_dim = hidden_dim // 2
        self.precision = precision
        self.register_buffer("packed_weights",
                             torch.zeros(num_experts, ffn_dim, self.packed_dim, dtype=torch.uint8))
        self.register_buffer("scales", torch.zeros(num_experts, ffn_dim, dtype=torch.float32))
        self.register_buffer("zero_points", torch.zeros(num_experts, ffn_dim, dtype=torch.float32))

    @classmethod
    def from_weights(cls, weights_list, precision: str = "default"):
        module = cls(len(weights_list), weights_list[0].shape[1], weights_list[0].shape[0], precision=precision)
        packed, scales, zero_points = quantize_weights_moe(weights_list)
        module.packed_weights = packed
        module.scales = scales
        module.zero_points = zero_points
        return module

    def forward(self, inputs, expert_ids, tokens_per_expert, input_offsets):
        """inputs ``[T, K]`` float32 with rows grouped by expert -> ``[T, N]`` float32."""
        if not CUDA_AVAILABLE:
            raise RuntimeError("CUDA kernel not available")
        from . import ops
        return ops.moe_forward(self.packed_weights, self.scales, self.zero_points, inputs, expert_ids,
                               tokens_per_expert, input_offsets, precision=self.precision)


class QuantizedMoEExpert(nn.Module):
    def __init__(self, in_features: int, out_features: int, precision: str = "default"):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.precision = precision
        self.register_buffer("packed_weights", torch.zeros(out_features, in_features // 2, dtype=torch.uint8))
        self.register_buffer("scales", torch.zeros(out_features, dtype=torch.float32))
        self.register_buffer("zero_points", torch.zeros(out_features, dtype=torch.float32))

    @classmethod
    def from_fp16(cls, weight: torch.Tensor, precision: str = "default") -> "QuantizedMoEExpert":
        assert weight.shape[1] % 2 == 0, "in_features must be even for INT4 packing"
        expert = cls(weight.shape[1], weight.shape[0], precision=precision)
        packed, scales, zero_points = quantize_weights(weight.float())
        expert.packed_weights = packed
        expert.scales = scales
        expert.zero_points = zero_points
        return expert

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape[0] == 0:      # reference quirk: always float16 (moe_int4_module.py:65-68)
            return torch.empty(0, self.out_features, device=x.device, dtype=torch.float16)
        if x.is_cuda:
            from . import ops
            return ops.linear_forward_any(x.contiguous(), self.packed_weights, self.scales, self.zero_points,
                                          precision=self.precision)          # in x's dtype (:70-72)
        w = dequantize_weights(self.packed_weights, self.scales, self.zero_points)
        return x @ w.T.to(x.dtype)

    @property
    def weight_memory_bytes(self) -> int:
        return self.packed_weights.numel() + 4 * self.scales.numel() + 4 * self.zero_points.numel()


class QuantizedMoE(nn.Module):
    def __init__(self, num_experts: int, hidden_dim: int, ffn_dim: int, precision: str = "default"):
        super().__init__()
        self.num_experts = num_experts
        self.hidden_dim = hidden_dim
        self.ffn_dim = ffn_dim
        self.precision = precision
        self.experts = nn.ModuleList([QuantizedMoEExpert(hidden_dim, ffn_dim, precision)
                                      for _ in range(num_experts)])
        self._stacked = None        # lazily built [E,N,K/2] view of the experts' buffers for the grouped launch

    @classmethod
    def from_fp16_weights(cls, weights: List[torch.Tensor], precision: str = "default") -> "QuantizedMoE":
        assert len(weights) > 0
        moe = cls(len(weights), weights[0].shape[1], weights[0].shape[0], precision=precision)
        for i, w in enumerate(weights):
            moe.experts[i] = QuantizedMoEExpert.from_fp16(w, precision=precision)
        return moe

    def _stack(self, device):
        """The experts' buffers as ONE [E, N, K/2] / [E, N] / [E, N] set for the grouped launch -- without a second copy of
        the weights: the stacked tensors become the storage and every expert's registered buffers are re-bound to views
        of them (state_dict keys, shapes and in-place loads are unchanged; ``.to()`` / a replaced expert break the
        views, which the next call detects by address and repairs).  Only when the module lives on another device than
        the inputs is a side copy kept, as before."""
        names = ("packed_weights", "scales", "zero_points")
        st = self._stacked
        if st is not None and st[0].device == device:
            views = all(getattr(e, n).data_ptr() == st[k][i].data_ptr() for i, e in enumerate(self.experts) for k, n in enumerate(names))
            if views:                                            # the experts' buffers ARE the stacked storage
                return st[:3]
            key = tuple((getattr(e, n).data_ptr(), getattr(e, n)._version) for e in self.experts for n in names)
            if st[3] == key:                                     # side copy of a module on another device, still current
                return st[:3]
        on_device = all(getattr(e, n).device == device for e in self.experts for n in names)
        stacked = tuple(torch.stack([getattr(e, n) for e in self.experts]).to(device) for n in names)
        key = None
        if on_device:
            for i, e in enumerate(self.experts):
                for k, n in enumerate(names):
                    e._buffers[n] = stacked[k][i]               # a view: the per-expert copy is released
        else:
            key = tuple((getattr(e, n).data_ptr(), getattr(e, n)._version) for e in self.experts for n in names)
        self._stacked = stacked + (key,)
        return stacked

    def forward(self, expert_inputs: List[torch.Tensor]) -> List[torch.Tensor]:
        """One tensor ``[m_e, K]`` per expert in, one ``[m_e, N]`` per expert out (in x's dtype;
        empty inputs give an empty float16 tensor, as the reference does)."""
        if len(expert_inputs) == 0 or not expert_inputs[0].is_cuda:
            return [expert(x) for expert, x in zip(self.experts, expert_inputs)]
        from . import ops
        dev = expert_inputs[0].device
        counts = [int(x.shape[0]) for x in expert_inputs]
        packed, scales, zps = self._stack(dev)
        dtypes = {x.dtype for x in expert_inputs}
        common = dtypes.pop() if len(dtypes) == 1 else torch.float32      # mixed dtypes: compute once in float32
        if common not in (torch.float32, torch.float16, torch.bfloat16):
            common = torch.float32
        grouped = torch.cat([x.to(common) for x in expert_inputs], dim=0).contiguous()
        tpe = torch.tensor(counts, dtype=torch.int32)
        offs = torch.cumsum(tpe, 0, dtype=torch.int32) - tpe
        out = ops.moe_forward_any(packed, scales, zps, grouped, None, tpe.to(dev, non_blocking=True),
                                  offs.to(dev, non_blocking=True), precision=self.precision) \
            if grouped.shape[0] else grouped.new_zeros((0, self.ffn_dim))
        outs, o = [], 0
        for x, c in zip(expert_inputs, counts):
            if c == 0:
                outs.append(torch.empty(0, self.ffn_dim, device=x.device, dtype=torch.float16))
            else:
                outs.append(out[o:o + c].to(x.dtype))
            o += c
        return outs

    @property
    def total_memory_bytes(self) -> int:
        return sum(e.weight_memory_bytes for e in self.experts)


class QuantizedMoEFFN(nn.Module):
    """Full gated FFN experts in INT4: ``down( silu(gate(x)) * up(x) )`` per expert, rows
    pre-grouped by expert.  The reference models only the up projection ("just the up projection for
    simplicity", benchmark/moe_grouped_gemm/config.py:50-52); this is the step either side of it in a
    Mixtral / DeepSeek block, built from the same kernels:

      * gate and up are ONE grouped GEMM over the stacked ``[E, 2F, H]`` weights (one pass over x),
      * silu(gate) * up is fused into the down GEMM's activation pre-pass (``fql_moe_gated_fwd_f32``),
        so the ``[T, F]`` hidden activation never exists in memory.

    Weights are quantised per output row with ``quantize_weights`` (per-row scale / zero point, as
    ``QuantizedMoEExpert.from_fp16`` does)."""

    def __init__(self, num_experts: int, hidden_dim: int, ffn_dim: int, precision: str = "default"):
        super().__init__()
        assert hidden_dim % 32 == 0 and ffn_dim % 32 == 0, "hidden_dim and ffn_dim must be multiples of 32"
        self.num_experts, self.hidden_dim, self.ffn_dim, self.precision = num_experts, hidden_dim, ffn_dim, precision
        E, H, F = num_experts, hidden_dim, ffn_dim
        self.register_buffer("gate_up_packed", torch.zeros(E, 2 * F, H // 2, dtype=torch.uint8))
        self.register_buffer("gate_up_scales", torch.zeros(E, 2 * F, dtype=torch.float32))
        self.register_buffer("gate_up_zero_points", torch.zeros(E, 2 * F, dtype=torch.float32))
        self.register_buffer("down_packed", torch.zeros(E, H, F // 2, dtype=torch.uint8))
        self.register_buffer("down_scales", torch.zeros(E, H, dtype=torch.float32))
        self.register_buffer("down_zero_points", torch.zeros(E, H, dtype=torch.float32))

    @classmethod
    def from_weights(cls, gate: List[torch.Tensor], up: List[torch.Tensor], down: List[torch.Tensor],
                     precision: str = "default") -> "QuantizedMoEFFN":
        """``gate[e]``, ``up[e]``: ``[F, H]``; ``down[e]``: ``[H, F]`` (nn.Linear weight layout)."""
        E = len(gate)
        F, H = gate[0].shape
        m = cls(E, H, F, precision)
        gu = [quantize_weights(torch.cat([g.float(), u.float()], dim=0)) for g, u in zip(gate, up)]
        dn = [quantize_weights(d.float()) for d in down]
        m.gate_up_packed = torch.stack([t[0] for t in gu])
        m.gate_up_scales = torch.stack([t[1] for t in gu])
        m.gate_up_zero_points = torch.stack([t[2] for t in gu])
        m.down_packed = torch.stack([t[0] for t in dn])
        m.down_scales = torch.stack([t[1] for t in dn])
        m.down_zero_points = torch.stack([t[2] for t in dn])
        return m

    def forward(self, inputs, tokens_per_expert, input_offsets):
        """inputs ``[T, H]`` float32 rows grouped by expert -> ``[T, H]`` float32."""
        if not inputs.is_cuda:
            raise RuntimeError("QuantizedMoEFFN runs on the GPU (the product path has no CPU fallback)")
        from . import ops
        gate_up = ops.moe_forward(self.gate_up_packed, self.gate_up_scales, self.gate_up_zero_points, inputs, None,
                                  tokens_per_expert, input_offsets, precision=self.precision)
        return ops.moe_gated_forward(self.down_packed, self.down_scales, self.down_zero_points, gate_up,
                                     tokens_per_expert, input_offsets, precision=self.precision)

    @property
    def total_memory_bytes(self) -> int:
        return sum(b.numel() * b.element_size() for b in self.buffers())
