"""Operator boundary: tensor checks + raw-pointer calls into libfql_int4.so.

``linear_forward`` mirrors ``fused_quant_linear_cuda.forward`` (reference
csrc/quantized_linear.cpp:22-28 and the checks of csrc/quantized_linear_kernel.cu:293-378);
``moe_forward`` mirrors ``moe_int4_cuda.forward`` (reference csrc/moe_int4_kernel.cu:93-141).
Outputs are allocated here with torch (``torch.empty``), launches go to torch's current
stream, nothing synchronises.  GPU tensors only -- a CPU tensor raises, exactly like the
reference's TORCH_CHECK(input.is_cuda()).
"""
from __future__ import annotations

import torch

from . import _native

_PRECISIONS = {"default": _native.PRECISION_DEFAULT, "exact": _native.PRECISION_EXACT,
               "fast": _native.PRECISION_FAST, "int8": _native.PRECISION_INT8, "fp8": _native.PRECISION_FP8,
               0: 0, 1: 1, 2: 2, 3: 3, 8: 8}


def _planes(prec):
    """Byte planes of the activation workspace: 3 / 2 / 1 int8 limbs, or one plane of e4m3 values."""
    return {0: 3, 8: 1}.get(prec, prec)


def _precision(p):
    try:
        return _PRECISIONS[p]
    except KeyError:
        raise ValueError(f"precision must be 'default', 'exact', 'fast', 'int8' or 'fp8', got {p!r}") from None


def _stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


def _workspace(nbytes, device):
    if nbytes == 0:
        return None, 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return ws, ws.data_ptr()


def linear_forward(input, packed_weights, scales, zero_points, precision="default", bias=None):
    """Fused 4-bit dequantize + linear forward.  input [K] or [B,K] float32 -> [N] or [B,N].
    ``bias`` [N] float32 (optional, not in the reference: python/module.py:84) is added in the kernels' epilogues."""
    squeeze = False
    if input.dim() == 1:                                   # :302-306
        input = input.unsqueeze(0)
        squeeze = True
    # device / layout / dtype checks, same order and wording as :311-335
    if not input.is_cuda:
        raise RuntimeError("input must be a CUDA tensor")
    if not packed_weights.is_cuda:
        raise RuntimeError("packed_weights must be a CUDA tensor")
    if not scales.is_cuda:
        raise RuntimeError("scales must be a CUDA tensor")
    if not zero_points.is_cuda:
        raise RuntimeError("zero_points must be a CUDA tensor")
    if not input.is_contiguous():
        raise RuntimeError("input must be contiguous")
    if not packed_weights.is_contiguous():
        raise RuntimeError("packed_weights must be contiguous")
    if input.dtype != torch.float32:
        raise RuntimeError("input must be float32")
    if packed_weights.dtype != torch.uint8:
        raise RuntimeError("packed_weights must be uint8")
    if scales.dtype != torch.float32:
        raise RuntimeError("scales must be float32")
    if zero_points.dtype != torch.float32:
        raise RuntimeError("zero_points must be float32")
    if input.dim() != 2 or packed_weights.dim() != 2:
        raise RuntimeError("input must be 1-D or 2-D and packed_weights 2-D")
    B, K = input.shape
    N, packed_dim = packed_weights.shape
    if packed_dim != K // 2 or K % 2 != 0:
        raise RuntimeError("packed_weights dim 1 must be input_dim / 2")
    if scales.dim() == 2 and scales.shape[1] > 1:           # per-group scales along K (float32 contraction: include/fql_int4.h)
        out = _linear_group_forward(input, packed_weights, scales, zero_points, bias, precision)
        return out.squeeze(0) if squeeze else out
    # not checked by the reference (silent UB there); checked here
    if scales.numel() != N or zero_points.numel() != N:
        raise RuntimeError("scales and zero_points must have output_dim elements")
    dev = input.device
    if packed_weights.device != dev or scales.device != dev or zero_points.device != dev:
        raise RuntimeError("all tensors must be on the same device")
    scales = scales.contiguous()
    zero_points = zero_points.contiguous()
    if bias is not None:
        if not bias.is_cuda or bias.device != dev or bias.dtype != torch.float32 or bias.numel() != N:
            raise RuntimeError("bias must be a float32 tensor with output_dim elements on the input's device")
        bias = bias.contiguous()

    L = _native.lib()
    prec = _precision(precision)
    out = torch.empty((B, N), dtype=torch.float32, device=dev)     # :338
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_linear_workspace_bytes(B, K, N, prec), dev)
        if bias is None:
            rc = L.fql_linear_fwd_f32(input.data_ptr(), packed_weights.data_ptr(), scales.data_ptr(),
                                      zero_points.data_ptr(), out.data_ptr(), B, K, N, prec,
                                      ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
        else:
            rc = L.fql_linear_bias_fwd_f32(input.data_ptr(), packed_weights.data_ptr(), scales.data_ptr(),
                                           zero_points.data_ptr(), bias.data_ptr(), out.data_ptr(), B, K, N, prec,
                                           ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_linear_fwd_f32")
    return out.squeeze(0) if squeeze else out                      # :373-375


def _linear_group_forward(input, packed_weights, scales, zero_points, bias, precision="default"):
    """Per-group scales along K (``scales`` / ``zero_points`` [N, K / group_size]): fql_linear_group_fwd_f32."""
    B, K = input.shape
    N = packed_weights.shape[0]
    if tuple(scales.shape) != tuple(zero_points.shape) or scales.shape[0] != N or K % scales.shape[1] != 0:
        raise RuntimeError("per-group scales and zero_points must be [output_dim, input_dim / group_size]")
    group = K // scales.shape[1]
    if group % 2 != 0:
        raise RuntimeError("group_size must be even")
    dev = input.device
    for t in (packed_weights, scales, zero_points):
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    if packed_weights.dtype != torch.uint8 or scales.dtype != torch.float32 or zero_points.dtype != torch.float32:
        raise RuntimeError("packed_weights must be uint8, scales and zero_points float32")
    if bias is not None:
        if bias.device != dev or bias.dtype != torch.float32 or bias.numel() != N:
            raise RuntimeError("bias must be a float32 tensor with output_dim elements on the input's device")
        bias = bias.contiguous()
    # (named: a temporary made by .contiguous() must outlive the launch that reads it)
    x_c, p_c, s_c, z_c = input.contiguous(), packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()
    out = torch.empty((B, N), dtype=torch.float32, device=dev)
    prec = _precision(precision)
    with torch.cuda.device(dev):
        L = _native.lib()
        ws, ws_ptr = _workspace(L.fql_group_workspace_bytes(1, B, K, N, group, prec), dev)
        rc = L.fql_linear_group_ws_fwd_f32(x_c.data_ptr(), p_c.data_ptr(), s_c.data_ptr(), z_c.data_ptr(),
                                           None if bias is None else bias.data_ptr(), out.data_ptr(), B, K, N,
                                           group, prec, ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_linear_group_ws_fwd_f32")
    return out


def _check_group_scales(packed_weights, scales, zero_points, dev, E, N, K):
    """Per-group constants of a grouped call: CUDA tensors on ``dev``, uint8 / float32, [E, N, K / group_size] with an even
    group size.  Returns the group size."""
    for name, t in (("packed_weights", packed_weights), ("scales", scales), ("zero_points", zero_points)):
        if not t.is_cuda or t.device != dev:
            raise RuntimeError(f"{name} must be a CUDA tensor on the inputs' device")
    if packed_weights.dtype != torch.uint8:
        raise RuntimeError("packed_weights must be uint8")
    if scales.dtype != torch.float32 or zero_points.dtype != torch.float32:
        raise RuntimeError("scales and zero_points must be float32")
    if scales.dim() != 3 or tuple(scales.shape[:2]) != (E, N) or tuple(zero_points.shape) != tuple(scales.shape) \
            or scales.shape[2] == 0 or K % scales.shape[2] != 0:
        raise RuntimeError("per-group scales and zero_points must be [E, N, K / group_size]")
    group = K // scales.shape[2]
    if group % 2 != 0:
        raise RuntimeError("group_size must be even")
    return group


def moe_group_forward(packed_weights, scales, zero_points, inputs, tokens_per_expert, input_offsets, precision="default"):
    """Grouped per-expert INT4 GEMM with per-GROUP scales along K: ``scales`` / ``zero_points`` [E, N, K / group_size].
    Batches of 8+ rows per expert run on the INT8 matrix cores (limb accumulators folded in float32 per group), smaller
    ones on the float32 matrix-core / FMA kernels (include/fql_int4.h); rows no expert covers are zero."""
    if not inputs.is_cuda or inputs.dtype != torch.float32 or inputs.dim() != 2 or packed_weights.dim() != 3:
        raise RuntimeError("inputs must be a CUDA float32 [T, K] tensor and packed_weights [E, N, K/2]")
    E, N, K2 = packed_weights.shape
    T, K = inputs.shape
    if K != 2 * K2:
        raise RuntimeError("packed_weights dim 2 must be hidden_dim / 2")
    dev = inputs.device
    group = _check_group_scales(packed_weights, scales, zero_points, dev, E, N, K)
    for name, t in (("tokens_per_expert", tokens_per_expert), ("input_offsets", input_offsets)):
        if not t.is_cuda or t.device != dev:
            raise RuntimeError(f"{name} must be a CUDA tensor on the inputs' device")
        if t.numel() != E:
            raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    # (named: a temporary made by .contiguous() must outlive the launch that reads it)
    x_c, p_c, s_c, z_c = inputs.contiguous(), packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()
    out = torch.empty((T, N), dtype=torch.float32, device=dev)
    prec = _precision(precision)
    with torch.cuda.device(dev):
        L = _native.lib()
        ws, ws_ptr = _workspace(L.fql_group_workspace_bytes(E, T, K, N, group, prec), dev)
        rc = L.fql_moe_group_ws_fwd_f32(p_c.data_ptr(), s_c.data_ptr(), z_c.data_ptr(), x_c.data_ptr(),
                                        tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, T, K, N, group, prec,
                                        ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_moe_group_ws_fwd_f32")
    return out


def moe_forward(packed_weights, scales, zero_points, inputs, expert_ids, tokens_per_expert,
                input_offsets, precision="default"):
    """Grouped per-expert INT4 GEMM over rows pre-grouped by expert.

    packed_weights [E,N,K/2] u8, scales/zero_points [E,N] f32, inputs [T,K] f32,
    tokens_per_expert / input_offsets [E] int32 on the device (consumed there, no .item()).
    ``expert_ids`` is accepted and ignored, as in the reference (csrc/moe_int4_kernel.cu:98).
    Returns [T,N] float32; rows covered by no expert are zero (reference: torch::zeros :109).
    """
    del expert_ids
    for name, t in (("packed_weights", packed_weights), ("scales", scales), ("zero_points", zero_points),
                    ("inputs", inputs), ("tokens_per_expert", tokens_per_expert),
                    ("input_offsets", input_offsets)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
    if packed_weights.dtype != torch.uint8 or packed_weights.dim() != 3:
        raise RuntimeError("packed_weights must be uint8 [num_experts, ffn_dim, hidden_dim/2]")
    if inputs.dtype != torch.float32 or inputs.dim() != 2:
        raise RuntimeError("inputs must be float32 [total_tokens, hidden_dim]")
    if scales.dtype != torch.float32 or zero_points.dtype != torch.float32:
        raise RuntimeError("scales and zero_points must be float32")
    E, N, packed_dim = packed_weights.shape
    T, K = inputs.shape
    if K % 2 != 0 or packed_dim != K // 2:
        raise RuntimeError("packed_weights dim 2 must be hidden_dim / 2")
    if tuple(scales.shape) != (E, N) or tuple(zero_points.shape) != (E, N):
        raise RuntimeError("scales and zero_points must be [num_experts, ffn_dim]")
    if tokens_per_expert.numel() != E or input_offsets.numel() != E:
        raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    dev = inputs.device
    for t in (packed_weights, scales, zero_points, tokens_per_expert, input_offsets):
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    packed_weights = packed_weights.contiguous()
    inputs = inputs.contiguous()
    scales = scales.contiguous()
    zero_points = zero_points.contiguous()

    L = _native.lib()
    prec = _precision(precision)
    out = torch.empty((T, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_moe_workspace_bytes(E, T, K, N, prec), dev)
        rc = L.fql_moe_fwd_f32(packed_weights.data_ptr(), scales.data_ptr(), zero_points.data_ptr(),
                               inputs.data_ptr(), tpe.data_ptr(), offs.data_ptr(), out.data_ptr(),
                               E, T, K, N, prec, ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_moe_fwd_f32")
    return out


_DTYPES = {torch.float32: _native.DTYPE_F32, torch.float16: _native.DTYPE_F16, torch.bfloat16: _native.DTYPE_BF16}


def linear_forward_any(input, packed_weights, scales, zero_points, precision="default", out_dtype=None):
    """``linear_forward`` for float32 / float16 / bfloat16 activations, output in ``out_dtype`` (default: the
    input's).  Equal bit for bit to ``linear_forward(input.float()).to(out_dtype)``; on the MFMA path the two
    conversion passes are fused into the kernels, elsewhere torch converts."""
    out_dtype = input.dtype if out_dtype is None else out_dtype
    if input.dtype not in _DTYPES or out_dtype not in _DTYPES:
        raise RuntimeError("activations and outputs must be float32, float16 or bfloat16")
    if input.dtype == torch.float32 and out_dtype == torch.float32:
        return linear_forward(input, packed_weights, scales, zero_points, precision=precision)
    x2 = input.unsqueeze(0) if input.dim() == 1 else input
    prec = _precision(precision)
    L = _native.lib()
    native = (x2.is_cuda and x2.dim() == 2 and x2.is_contiguous() and packed_weights.is_cuda
              and packed_weights.dtype == torch.uint8 and packed_weights.dim() == 2 and packed_weights.is_contiguous()
              and packed_weights.shape[1] * 2 == x2.shape[1]
              and L.fql_native_dtype_supported(x2.shape[0], 1, x2.shape[1], packed_weights.shape[0], prec,
                                               packed_weights.data_ptr(), 0) == 1)
    if not native:
        return linear_forward(input.float().contiguous(), packed_weights, scales, zero_points,
                              precision=precision).to(out_dtype)
    B, K = x2.shape
    N = packed_weights.shape[0]
    dev = x2.device
    if scales.numel() != N or zero_points.numel() != N or scales.dtype != torch.float32 or zero_points.dtype != torch.float32:
        raise RuntimeError("scales and zero_points must be float32 with output_dim elements")
    scales, zero_points = scales.contiguous(), zero_points.contiguous()
    out = torch.empty((B, N), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_linear_workspace_bytes(B, K, N, prec), dev)
        rc = L.fql_linear_fwd(x2.data_ptr(), _DTYPES[x2.dtype], packed_weights.data_ptr(), scales.data_ptr(),
                              zero_points.data_ptr(), out.data_ptr(), _DTYPES[out_dtype], B, K, N, prec,
                              ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_linear_fwd")
    return out.squeeze(0) if input.dim() == 1 else out


def moe_forward_any(packed_weights, scales, zero_points, inputs, expert_ids, tokens_per_expert, input_offsets,
                    precision="default", out_dtype=None):
    """``moe_forward`` for float32 / float16 / bfloat16 rows (the reference's MoE benches feed float16,
    benchmark/moe_grouped_gemm/moe_int4_module.py:161-165), output in ``out_dtype`` (default: the input's).
    Equal bit for bit to ``moe_forward(inputs.float()).to(out_dtype)``."""
    out_dtype = inputs.dtype if out_dtype is None else out_dtype
    if inputs.dtype not in _DTYPES or out_dtype not in _DTYPES:
        raise RuntimeError("activations and outputs must be float32, float16 or bfloat16")
    if inputs.dtype == torch.float32 and out_dtype == torch.float32:
        return moe_forward(packed_weights, scales, zero_points, inputs, expert_ids, tokens_per_expert, input_offsets,
                           precision=precision)
    prec = _precision(precision)
    L = _native.lib()
    ok = (inputs.is_cuda and inputs.dim() == 2 and packed_weights.is_cuda and packed_weights.dim() == 3
          and packed_weights.dtype == torch.uint8 and packed_weights.shape[2] * 2 == inputs.shape[1])
    native = ok and L.fql_native_dtype_supported(inputs.shape[0], packed_weights.shape[0], inputs.shape[1],
                                                 packed_weights.shape[1], prec, packed_weights.data_ptr(), 1) == 1
    if not native:
        return moe_forward(packed_weights, scales, zero_points, inputs.float().contiguous(), expert_ids,
                           tokens_per_expert, input_offsets, precision=precision).to(out_dtype)
    E, N, _ = packed_weights.shape
    T, K = inputs.shape
    if tuple(scales.shape) != (E, N) or tuple(zero_points.shape) != (E, N):
        raise RuntimeError("scales and zero_points must be [num_experts, ffn_dim]")
    if tokens_per_expert.numel() != E or input_offsets.numel() != E:
        raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    dev = inputs.device
    inputs = inputs.contiguous()
    packed_weights, scales, zero_points = packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty((T, N), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_moe_workspace_bytes(E, T, K, N, prec), dev)
        rc = L.fql_moe_fwd(packed_weights.data_ptr(), scales.data_ptr(), zero_points.data_ptr(), inputs.data_ptr(),
                           _DTYPES[inputs.dtype], tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), _DTYPES[out_dtype],
                           E, T, K, N, prec, ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_moe_fwd")
    return out


def _check_grouped_weights(packed_weights, scales, zero_points, dev, K):
    """Shape / dtype / device checks shared by the grouped wrappers: a wrong-shaped or foreign-device tensor must become a
    Python error here, not an out-of-bounds read inside a kernel."""
    if not packed_weights.is_cuda or packed_weights.dtype != torch.uint8 or packed_weights.dim() != 3:
        raise RuntimeError("packed_weights must be a CUDA uint8 [num_experts, ffn_dim, hidden_dim/2] tensor")
    E, N, packed_dim = packed_weights.shape
    if packed_dim * 2 != K:
        raise RuntimeError("packed_weights dim 2 must be hidden_dim / 2")
    for name, t in (("scales", scales), ("zero_points", zero_points)):
        if not t.is_cuda or t.dtype != torch.float32 or tuple(t.shape) != (E, N):
            raise RuntimeError(f"{name} must be a CUDA float32 [num_experts, ffn_dim] tensor")
    for t in (packed_weights, scales, zero_points):
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device")
    return E, N


def moe_gather_forward(packed_weights, scales, zero_points, tokens, row_index, tokens_per_expert,
                       input_offsets, precision="default", row_weight=None):
    """Grouped per-expert INT4 GEMM with the dispatch gather fused in: grouped row t = tokens[row_index[t]].
    ``row_weight`` [T] float32 (optional): grouped row t of the result is multiplied by row_weight[t] in the GEMM epilogue
    (the routing weight of its (token, slot) pair), so that ``combine(y, pos, None)`` is a pure gather-add.

    ``tokens`` [n_tokens, K] float32 in token order, ``row_index`` [T] int32 (e.g. from
    ``routing.dispatch_indices``).  Returns the grouped outputs [T, N]; un-sort / combine with
    ``routing.combine_grouped``.  The [T, K] gathered activations are never materialised."""
    for name, t in (("packed_weights", packed_weights), ("tokens", tokens), ("row_index", row_index),
                    ("tokens_per_expert", tokens_per_expert), ("input_offsets", input_offsets)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
    if tokens.dtype != torch.float32 or tokens.dim() != 2:
        raise RuntimeError("tokens must be float32 [n_tokens, hidden_dim]")
    n_tokens, K = tokens.shape
    dev = tokens.device
    E, N = _check_grouped_weights(packed_weights, scales, zero_points, dev, K)
    if K % 32 != 0:
        raise RuntimeError("fused gather needs hidden_dim % 32 == 0")
    if tokens_per_expert.numel() != E or input_offsets.numel() != E:
        raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    T = row_index.numel()
    ri = row_index.to(device=dev, dtype=torch.int32).contiguous()
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    tokens = tokens.contiguous()
    if row_weight is not None:
        if not row_weight.is_cuda or row_weight.device != dev or row_weight.numel() != T:
            raise RuntimeError("row_weight must be a CUDA tensor with one element per grouped row")
        row_weight = row_weight.to(torch.float32).contiguous()
    L = _native.lib()
    prec = _precision(precision)
    out = torch.empty((T, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_moe_workspace_bytes(E, T, K, N, prec), dev)
        packed_weights_c, scales_c, zero_points_c = packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()   # (named: must outlive the launch)
        if row_weight is None:
            rc = L.fql_moe_gather_fwd_f32(packed_weights_c.data_ptr(), scales_c.data_ptr(),
                                          zero_points_c.data_ptr(), tokens.data_ptr(), ri.data_ptr(), n_tokens,
                                          tpe.data_ptr(), offs.data_ptr(), out.data_ptr(), E, T, K, N, prec,
                                          ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
        else:
            rc = L.fql_moe_gather_scaled_fwd_f32(packed_weights_c.data_ptr(), scales_c.data_ptr(),
                                                 zero_points_c.data_ptr(), tokens.data_ptr(), ri.data_ptr(), n_tokens,
                                                 row_weight.data_ptr(), tpe.data_ptr(), offs.data_ptr(), out.data_ptr(),
                                                 E, T, K, N, prec, ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_moe_gather_fwd_f32")
    return out


def moe_gated_forward(packed_weights, scales, zero_points, gate_up, tokens_per_expert, input_offsets,
                      precision="default"):
    """Second GEMM of a gated FFN expert with the activation fused into its pre-pass:
    ``out[t] = W_e @ (silu(gate_up[t, :K]) * gate_up[t, K:])``; ``gate_up`` [T, 2K] float32 (the output of the
    fused gate|up projection), ``packed_weights`` [E, N, K/2].  The [T, K] hidden activation is never written."""
    if not gate_up.is_cuda or gate_up.dtype != torch.float32 or gate_up.dim() != 2:
        raise RuntimeError("gate_up must be a CUDA float32 [T, 2K] tensor")
    T, K2 = gate_up.shape
    K = K2 // 2
    dev = gate_up.device
    if K2 % 2 or K % 32:
        raise RuntimeError("gate_up must be [T, 2K] with K % 32 == 0")
    E, N = _check_grouped_weights(packed_weights, scales, zero_points, dev, K)
    if tokens_per_expert.numel() != E or input_offsets.numel() != E:
        raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    gate_up = gate_up.contiguous()
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    L = _native.lib()
    prec = _precision(precision)
    out = torch.empty((T, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_moe_workspace_bytes(E, T, K, N, prec), dev)
        packed_weights_c, scales_c, zero_points_c = packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()   # (named: must outlive the launch)
        rc = L.fql_moe_gated_fwd_f32(packed_weights_c.data_ptr(), scales_c.data_ptr(),
                                     zero_points_c.data_ptr(), gate_up.data_ptr(), tpe.data_ptr(),
                                     offs.data_ptr(), out.data_ptr(), E, T, K, N, prec,
                                     ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_moe_gated_fwd_f32")
    return out


ROUTE_MAX_EXPERTS = 128


def route_plan(expert_indices, num_experts):
    """Stable sort of the (token, slot) pairs by expert in ONE launch (replaces argsort + bincount + cumsum +
    index ops of routing.py:117-149).  ``expert_indices`` [T, top_k] integer, on the GPU.  Returns int32 device
    tensors ``(tokens_per_expert [E], input_offsets [E], token_of_sorted [T*top_k], pos_of_slot [T*top_k])``."""
    if not expert_indices.is_cuda or expert_indices.dim() != 2:
        raise RuntimeError("expert_indices must be a CUDA [tokens, top_k] tensor")
    if num_experts > ROUTE_MAX_EXPERTS:
        raise RuntimeError(f"route_plan supports up to {ROUTE_MAX_EXPERTS} experts")
    dev = expert_indices.device
    T, top_k = expert_indices.shape
    flat = expert_indices.reshape(-1).to(torch.int32).contiguous()
    n = flat.numel()
    counts = torch.empty(num_experts, dtype=torch.int32, device=dev)
    offsets = torch.empty(num_experts, dtype=torch.int32, device=dev)
    token_of_sorted = torch.empty(n, dtype=torch.int32, device=dev)
    pos_of_slot = torch.empty(n, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _native.lib().fql_route_plan_i32(flat.data_ptr(), n, top_k, num_experts, counts.data_ptr(),
                                              offsets.data_ptr(), token_of_sorted.data_ptr(), pos_of_slot.data_ptr(),
                                              _stream_ptr(dev))
    _native.check(rc, "fql_route_plan_i32")
    return counts, offsets, token_of_sorted, pos_of_slot


def combine(y, pos_of_slot, expert_weights, top_k=None):
    """out[t] = sum_k expert_weights[t, k] * y[pos_of_slot[t*top_k + k]] in one launch (routing.py:172-189).
    ``expert_weights=None`` (with ``top_k``): the rows already carry their weights -- a pure gather-add."""
    if not y.is_cuda or y.dtype != torch.float32 or y.dim() != 2:
        raise RuntimeError("y must be a CUDA float32 [rows, N] tensor")
    if expert_weights is None:
        if not top_k:
            raise RuntimeError("top_k is needed when the rows carry their weights")
        T = pos_of_slot.numel() // top_k
        if T > 65535:
            raise RuntimeError("combine handles up to 65535 tokens per call")
        dev = y.device
        y = y.contiguous()
        pos = pos_of_slot.to(device=dev, dtype=torch.int32).contiguous()
        out = torch.empty((T, y.shape[1]), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = _native.lib().fql_combine_f32(y.data_ptr(), pos.data_ptr(), None, out.data_ptr(), T, top_k,
                                               y.shape[1], y.shape[0], _stream_ptr(dev))
        _native.check(rc, "fql_combine_f32")
        return out
    T, top_k = expert_weights.shape
    if T > 65535:
        raise RuntimeError("combine handles up to 65535 tokens per call")
    dev = y.device
    y = y.contiguous()
    w = expert_weights.to(device=dev, dtype=torch.float32).contiguous()
    pos = pos_of_slot.to(device=dev, dtype=torch.int32).contiguous()
    out = torch.empty((T, y.shape[1]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = _native.lib().fql_combine_f32(y.data_ptr(), pos.data_ptr(), w.data_ptr(), out.data_ptr(), T, top_k,
                                           y.shape[1], y.shape[0], _stream_ptr(dev))
    _native.check(rc, "fql_combine_f32")
    return out


def regroup_index(recv_counts, total_rows):
    """Expert-parallel receive side: ``recv_counts`` [G, EL] (rows per source rank and local expert, in arrival
    order), ``total_rows`` = their sum (the caller knows it on the host from the all-to-all split sizes) ->
    ``(tokens_per_expert [EL], input_offsets [EL], gather [R], scatter [R])`` int32 device tensors, one launch."""
    if not recv_counts.is_cuda or recv_counts.dim() != 2:
        raise RuntimeError("recv_counts must be a CUDA [ranks, local_experts] tensor")
    dev = recv_counts.device
    G, EL = recv_counts.shape
    cnt = recv_counts.to(torch.int32).contiguous()
    tpe = torch.empty(EL, dtype=torch.int32, device=dev)
    offs = torch.empty(EL, dtype=torch.int32, device=dev)
    gather = torch.empty(max(total_rows, 1), dtype=torch.int32, device=dev)
    scatter = torch.empty(max(total_rows, 1), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _native.lib().fql_regroup_index_i32(cnt.data_ptr(), G, EL, tpe.data_ptr(), offs.data_ptr(),
                                                 gather.data_ptr(), scatter.data_ptr(), _stream_ptr(dev))
    _native.check(rc, "fql_regroup_index_i32")
    return tpe, offs, gather[:total_rows], scatter[:total_rows]


def quantize_rows(weight_fp32):
    """GPU quantize_weights (python/quantize.py:38-124), bit-exact with the host arithmetic."""
    if not weight_fp32.is_cuda or weight_fp32.dtype != torch.float32 or weight_fp32.dim() != 2:
        raise RuntimeError("weight must be a CUDA float32 [N,K] tensor")
    w = weight_fp32.contiguous()
    N, K = w.shape
    packed = torch.empty((N, K // 2), dtype=torch.uint8, device=w.device)
    scales = torch.empty((N,), dtype=torch.float32, device=w.device)
    zps = torch.empty((N,), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _native.lib().fql_quantize_rows_f32(w.data_ptr(), packed.data_ptr(), scales.data_ptr(), zps.data_ptr(),
                                                 N, K, _stream_ptr(w.device))
    _native.check(rc, "fql_quantize_rows_f32")
    return packed, scales, zps


def quantize_tensor(weight_fp32):
    """GPU per-tensor quantiser of one expert (python/moe_int4_module.py:45-76): scale / zp broadcast to [N]."""
    if not weight_fp32.is_cuda or weight_fp32.dtype != torch.float32 or weight_fp32.dim() != 2:
        raise RuntimeError("weight must be a CUDA float32 [N,K] tensor")
    w = weight_fp32.contiguous()
    N, K = w.shape
    packed = torch.empty((N, K // 2), dtype=torch.uint8, device=w.device)
    scales = torch.empty((N,), dtype=torch.float32, device=w.device)
    zps = torch.empty((N,), dtype=torch.float32, device=w.device)
    scratch = torch.empty((2 * N,), dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        rc = _native.lib().fql_quantize_tensor_f32(w.data_ptr(), packed.data_ptr(), scales.data_ptr(), zps.data_ptr(),
                                                   scratch.data_ptr(), N, K, _stream_ptr(w.device))
    _native.check(rc, "fql_quantize_tensor_f32")
    return packed, scales, zps


def unpack_nibbles(packed):
    """q[..., 2j] = packed[..., j] & 15; q[..., 2j+1] = packed[..., j] >> 4 on the device."""
    if not packed.is_cuda or packed.dtype != torch.uint8:
        raise RuntimeError("packed must be a CUDA uint8 tensor")
    packed = packed.contiguous()
    q = torch.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), dtype=torch.uint8, device=packed.device)
    with torch.cuda.device(packed.device):
        rc = _native.lib().fql_unpack_u8(packed.data_ptr(), q.data_ptr(), packed.numel(),
                                         _stream_ptr(packed.device))
    _native.check(rc, "fql_unpack_u8")
    return q


def dequantize_forward(packed_weights, scales, zero_points):
    """GPU dequantize_weights: [N,K/2] u8 -> [N,K] f32 (python/quantize.py:127-173)."""
    if not packed_weights.is_cuda:
        raise RuntimeError("packed_weights must be a CUDA tensor")
    packed_weights = packed_weights.contiguous()
    N, K2 = packed_weights.shape
    w = torch.empty((N, 2 * K2), dtype=torch.float32, device=packed_weights.device)
    with torch.cuda.device(packed_weights.device):
        scales_c, zero_points_c = scales.contiguous(), zero_points.contiguous()   # (named: must outlive the launch)
        rc = _native.lib().fql_dequantize_f32(packed_weights.data_ptr(), scales_c.data_ptr(),
                                              zero_points_c.data_ptr(), w.data_ptr(), N, 2 * K2,
                                              _stream_ptr(packed_weights.device))
    _native.check(rc, "fql_dequantize_f32")
    return w


def _sets(prec):
    """Limb sets of the two-phase buffers: 2 (main + residual of heavy-tailed rows) for 2 / 3 limbs, else 1."""
    return 2 if _planes(prec) >= 2 else 1


def act_quant(x, precision="default", tokens_per_expert=None, input_offsets=None, out=None):
    """Phase 1 of the MFMA path: float32 rows -> int8 limbs in MFMA-fragment order (+ delta, rowsum).
    Pass the device-side expert arrays for a grouped (MoE) layout, or neither for one group.
    Returns ``(limbs, delta [S, T], rowsum [S, limbs, T])`` with S = 2 sets for 2 / 3 limbs (set 1 = the residual of
    heavy-tailed rows, ``delta[1] == 0`` where a row has none) and S = 1 for int8 / fp8 (include/fql_int4.h)."""
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2:
        raise RuntimeError("x must be a CUDA float32 [T,K] tensor")
    x = x.contiguous()
    T, K = x.shape
    prec = _precision(precision)
    nl, ns = _planes(prec), _sets(prec)
    grouped = tokens_per_expert is not None
    E = tokens_per_expert.numel() if grouped else 1
    L = _native.lib()
    if out is not None:                 # reuse the buffers of an earlier call with the same shapes (timing loops)
        limbs, delta, rowsum = out
    else:
        limbs = torch.zeros(L.fql_act_limb_bytes(T, E, K, prec), dtype=torch.int8, device=x.device)
        delta = torch.zeros((ns, T), dtype=torch.float32, device=x.device)
        rowsum = torch.zeros((ns, nl, T), dtype=torch.int32, device=x.device)
    with torch.cuda.device(x.device):
        rc = L.fql_act_quant_f32(x.data_ptr(), limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(),
                                 tokens_per_expert.data_ptr() if grouped else None,
                                 input_offsets.data_ptr() if grouped else None, E, T, K, prec,
                                 _stream_ptr(x.device))
    _native.check(rc, "fql_act_quant_f32")
    return limbs, delta, rowsum


_SCRATCH = {}


def gemm_scratch(prec, device):
    """The residual-pass scratch of phase 2: one cached buffer per (device, STREAM) -- its slots are indexed by workgroup and
    wave only, so two launches that overlap on different streams must not share one; contents never outlive a launch."""
    n = _native.lib().fql_gemm_scratch_bytes(prec)
    if n == 0:
        return None
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, n)
    if key not in _SCRATCH:
        _SCRATCH[key] = torch.empty(n, dtype=torch.uint8, device=device)
    return _SCRATCH[key]


def gemm_i8(limbs, delta, rowsum, packed_weights, scales, zero_points, tokens_per_expert=None,
            input_offsets=None, precision="default", out=None):
    """Phase 2 of the MFMA path: grouped INT4 x INT8-limb GEMM over pre-converted activations
    (``act_quant`` output, produced with the same expert arrays).  ``packed_weights`` [N,K/2] (one
    group) or [E,N,K/2] with device-side ``tokens_per_expert`` / ``input_offsets``."""
    T = delta.shape[-1]
    grouped = packed_weights.dim() == 3
    E = packed_weights.shape[0] if grouped else 1
    N, K2 = packed_weights.shape[-2:]
    K = 2 * K2
    prec = _precision(precision)
    if _planes(prec) != rowsum.shape[-2] or _sets(prec) != delta.shape[0]:
        raise RuntimeError("limb count does not match precision")
    if limbs.numel() < _native.lib().fql_act_limb_bytes(T, E, K, prec):
        raise RuntimeError("limbs were not produced for this T, E, K")
    dev = limbs.device
    if out is None:
        out = torch.zeros((T, N), dtype=torch.float32, device=dev) if grouped else \
            torch.empty((T, N), dtype=torch.float32, device=dev)
    tpe_ptr = tokens_per_expert.data_ptr() if grouped else None
    off_ptr = input_offsets.data_ptr() if grouped else None
    scratch = gemm_scratch(prec, dev)
    with torch.cuda.device(dev):
        rc = _native.lib().fql_gemm_i8_f32(limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(),
                                           packed_weights.data_ptr(), scales.data_ptr(), zero_points.data_ptr(),
                                           tpe_ptr, off_ptr, out.data_ptr(), E, T, K, N, prec, _stream_ptr(dev),
                                           None if scratch is None else scratch.data_ptr(),
                                           0 if scratch is None else scratch.numel())
    _native.check(rc, "fql_gemm_i8_f32")
    return out


def tune_gemm_i8(cfg, limbs, delta, rowsum, packed_weights, scales, zero_points, tokens_per_expert, input_offsets, out,
                 E, T, K, N, precision):
    """Phase 2 with an explicit tile configuration id (tuning / test hook, not part of the public header)."""
    import ctypes
    fn = _native.lib().fql_tune_gemm_i8_f32
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] + [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    prec = _precision(precision)
    dev = limbs.device
    scratch = gemm_scratch(prec, dev)
    with torch.cuda.device(dev):
        return fn(cfg, limbs.data_ptr(), delta.data_ptr(), rowsum.data_ptr(), packed_weights.data_ptr(), scales.data_ptr(),
                  zero_points.data_ptr(), None if tokens_per_expert is None else tokens_per_expert.data_ptr(),
                  None if input_offsets is None else input_offsets.data_ptr(), out.data_ptr(), E, T, K, N, prec,
                  _stream_ptr(dev), None if scratch is None else scratch.data_ptr(), 0 if scratch is None else scratch.numel())


# ------------------------------------------------------------------------------------------------ fp8 activations
def _as_e4m3_bytes(t, name):
    if t.dtype == torch.uint8:
        return t
    if t.dtype == getattr(torch, "float8_e4m3fn", None):
        return t.view(torch.uint8)
    raise RuntimeError(f"{name} must be torch.float8_e4m3fn (or its uint8 bytes)")


def quantize_activations_fp8(x):
    """Per-row OCP e4m3 quantisation with torch ops (for callers that want to hold fp8 activations themselves):
    ``scale[t] = max|x[t]| / 448`` (1 for an all-zero row), ``x8 = (x / scale).to(torch.float8_e4m3fn)``.
    ``precision="fp8"`` on the float entry points does the same inside the fused pre-pass."""
    x = x.float()
    amax = x.abs().amax(dim=1)
    # tensor / tensor: a true float32 division (torch turns `/ 448.0` into a multiplication by the reciprocal, one ulp off)
    scale = torch.where(amax == 0, torch.ones_like(amax), amax / torch.full_like(amax, 448.0))
    return (x / scale[:, None]).to(torch.float8_e4m3fn), scale


def moe_forward_fp8(packed_weights, scales, zero_points, inputs_e4m3, act_scales, tokens_per_expert, input_offsets,
                    out_dtype=torch.float32):
    """Grouped per-expert INT4 GEMM over rows that are already fp8: ``inputs_e4m3`` [T, K] torch.float8_e4m3fn
    (or its uint8 bytes), ``act_scales`` [T] float32 or None.  One fp8 MFMA pass, float32 accumulation
    (BASELINE.json configs[4]).  Returns [T, N] in ``out_dtype``; rows no expert covers are zero."""
    x8 = _as_e4m3_bytes(inputs_e4m3, "inputs_e4m3")
    for name, t in (("packed_weights", packed_weights), ("scales", scales), ("zero_points", zero_points), ("inputs_e4m3", x8),
                    ("tokens_per_expert", tokens_per_expert), ("input_offsets", input_offsets)):
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a CUDA tensor")
    if packed_weights.dtype != torch.uint8 or packed_weights.dim() != 3 or x8.dim() != 2:
        raise RuntimeError("packed_weights must be uint8 [num_experts, ffn_dim, hidden_dim/2] and inputs [total_tokens, hidden_dim]")
    if out_dtype not in _DTYPES:
        raise RuntimeError("outputs must be float32, float16 or bfloat16")
    E, N, packed_dim = packed_weights.shape
    T, K = x8.shape
    if K % 32 != 0 or packed_dim != K // 2:
        raise RuntimeError("the fp8 path needs hidden_dim % 32 == 0 and packed_weights dim 2 == hidden_dim / 2")
    if tuple(scales.shape) != (E, N) or tuple(zero_points.shape) != (E, N) or scales.dtype != torch.float32 \
            or zero_points.dtype != torch.float32:
        raise RuntimeError("scales and zero_points must be float32 [num_experts, ffn_dim]")
    if tokens_per_expert.numel() != E or input_offsets.numel() != E:
        raise RuntimeError("tokens_per_expert and input_offsets must have num_experts elements")
    dev = x8.device
    if any(t.device != dev for t in (packed_weights, scales, zero_points)):
        raise RuntimeError("all tensors must be on the same device")
    if act_scales is not None:
        if act_scales.numel() != T:
            raise RuntimeError("act_scales must have one element per row")
        act_scales = act_scales.to(device=dev, dtype=torch.float32).contiguous()
    x8 = x8.contiguous()
    tpe = tokens_per_expert.to(device=dev, dtype=torch.int32).contiguous()
    offs = input_offsets.to(device=dev, dtype=torch.int32).contiguous()
    L = _native.lib()
    out = torch.empty((T, N), dtype=out_dtype, device=dev)
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_moe_workspace_bytes(E, T, K, N, _native.PRECISION_FP8), dev)
        packed_weights_c, scales_c, zero_points_c = packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()   # (named: must outlive the launch)
        rc = L.fql_moe_fwd_f8(packed_weights_c.data_ptr(), scales_c.data_ptr(),
                              zero_points_c.data_ptr(), x8.data_ptr(),
                              None if act_scales is None else act_scales.data_ptr(), tpe.data_ptr(), offs.data_ptr(),
                              out.data_ptr(), _DTYPES[out_dtype], E, T, K, N, ws_ptr, 0 if ws is None else ws.numel(),
                              _stream_ptr(dev))
    _native.check(rc, "fql_moe_fwd_f8")
    return out


def linear_forward_fp8(x_e4m3, act_scales, packed_weights, scales, zero_points, out_dtype=torch.float32):
    """Fused 4-bit dequantize + linear over fp8 rows: ``x_e4m3`` [B, K] torch.float8_e4m3fn (or uint8 bytes),
    ``act_scales`` [B] float32 or None -> [B, N] in ``out_dtype``."""
    x8 = _as_e4m3_bytes(x_e4m3, "x_e4m3")
    if not x8.is_cuda or not packed_weights.is_cuda:
        raise RuntimeError("x_e4m3 and packed_weights must be CUDA tensors")
    if x8.dim() != 2 or packed_weights.dim() != 2 or packed_weights.dtype != torch.uint8:
        raise RuntimeError("x_e4m3 must be [B, K] and packed_weights uint8 [N, K/2]")
    if out_dtype not in _DTYPES:
        raise RuntimeError("outputs must be float32, float16 or bfloat16")
    B, K = x8.shape
    N, packed_dim = packed_weights.shape
    if K % 32 != 0 or packed_dim != K // 2:
        raise RuntimeError("the fp8 path needs input_dim % 32 == 0 and packed_weights dim 1 == input_dim / 2")
    if scales.numel() != N or zero_points.numel() != N or scales.dtype != torch.float32 or zero_points.dtype != torch.float32:
        raise RuntimeError("scales and zero_points must be float32 with output_dim elements")
    dev = x8.device
    if any(t.device != dev for t in (packed_weights, scales, zero_points)):
        raise RuntimeError("all tensors must be on the same device")
    if act_scales is not None:
        if act_scales.numel() != B:
            raise RuntimeError("act_scales must have one element per row")
        act_scales = act_scales.to(device=dev, dtype=torch.float32).contiguous()
    x8 = x8.contiguous()
    L = _native.lib()
    out = torch.empty((B, N), dtype=out_dtype, device=dev)
    if B == 0:
        return out
    with torch.cuda.device(dev):
        ws, ws_ptr = _workspace(L.fql_linear_workspace_bytes(B, K, N, _native.PRECISION_FP8), dev)
        packed_weights_c, scales_c, zero_points_c = packed_weights.contiguous(), scales.contiguous(), zero_points.contiguous()   # (named: must outlive the launch)
        rc = L.fql_linear_fwd_f8(x8.data_ptr(), None if act_scales is None else act_scales.data_ptr(),
                                 packed_weights_c.data_ptr(), scales_c.data_ptr(),
                                 zero_points_c.data_ptr(), out.data_ptr(), _DTYPES[out_dtype], B, K, N,
                                 ws_ptr, 0 if ws is None else ws.numel(), _stream_ptr(dev))
    _native.check(rc, "fql_linear_fwd_f8")
    return out
