"""Token routing either side of the grouped GEMM: synthetic router, dispatch, combine.

Same functions and result types as the reference's benchmark/moe_grouped_gemm/routing.py:
``simulate_routing`` (:26-93), ``create_expert_inputs`` (:96-149), ``combine_expert_outputs``
(:152-189), ``RoutingResult`` (:15-23) -- plus ``dispatch_grouped`` which produces the
``(grouped_rows, tokens_per_expert, input_offsets)`` triple the fused MoE op consumes directly.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

import torch


@dataclass
class RoutingResult:
    expert_indices: torch.Tensor      # [num_tokens, top_k]
    expert_weights: torch.Tensor      # [num_tokens, top_k], renormalised softmax scores
    tokens_per_expert: List[int]
    expert_token_offsets: List[int]   # exclusive prefix sum of tokens_per_expert


def simulate_routing(num_tokens: int, num_experts: int, top_k: int, distribution: str = "skewed",
                     device: str = "cuda", seed: int = 42) -> RoutingResult:
    """Synthetic router: logits -> softmax -> top-k -> renormalise -> per-expert histogram.

    ``"skewed"``: log of a 1/(rank+1) prior plus N(0, 0.5) noise; ``"random"``: N(0,1) logits;
    ``"uniform"``: all-zero logits (degenerate: top-k ties send every token to the same experts).
    """
    torch.manual_seed(seed)
    if distribution == "uniform":
        logits = torch.zeros(num_tokens, num_experts, device=device)
    elif distribution == "skewed":
        prior = 1.0 / (torch.arange(num_experts, device=device, dtype=torch.float32) + 1)
        prior = prior / prior.sum()
        logits = torch.log(prior + 1e-10).unsqueeze(0).expand(num_tokens, -1)
        logits = logits + torch.randn_like(logits) * 0.5
    elif distribution == "random":
        logits = torch.randn(num_tokens, num_experts, device=device)
    else:
        raise ValueError(f"Unknown distribution: {distribution}")
    probs = torch.softmax(logits, dim=-1)
    expert_weights, expert_indices = torch.topk(probs, top_k, dim=-1)
    expert_weights = expert_weights / expert_weights.sum(dim=-1, keepdim=True)
    counts = torch.bincount(expert_indices.flatten().cpu(), minlength=num_experts).tolist()
    offsets = [0]
    for c in counts[:-1]:
        offsets.append(offsets[-1] + c)
    return RoutingResult(expert_indices, expert_weights, counts, offsets)


def balanced_routing(num_tokens: int, num_experts: int, top_k: int, device: str = "cuda",
                     seed: int = 42) -> RoutingResult:
    """Exactly ``num_tokens * top_k / num_experts`` rows per expert (the even split the reference's
    own MoE harness uses, python/moe_int4_module.py:198-206), distinct experts per token."""
    assert (num_tokens * top_k) % num_experts == 0 and top_k <= num_experts
    g = torch.Generator().manual_seed(seed)
    base = torch.arange(num_tokens * top_k) % num_experts
    idx = base.view(num_tokens, top_k)                      # consecutive experts: distinct within a token
    idx = idx[torch.randperm(num_tokens, generator=g)]
    w = torch.rand(num_tokens, top_k, generator=g) + 0.1
    w = w / w.sum(dim=-1, keepdim=True)
    counts = torch.bincount(idx.flatten(), minlength=num_experts).tolist()
    offsets = [0]
    for c in counts[:-1]:
        offsets.append(offsets[-1] + c)
    return RoutingResult(idx.to(device), w.to(device), counts, offsets)


def _plan_on_device(expert_indices: torch.Tensor, num_experts: int):
    """One-launch plan (csrc/fql_routing.h) when the indices live on the GPU; None otherwise."""
    if not expert_indices.is_cuda or expert_indices.numel() == 0:
        return None
    from . import ops
    if num_experts > ops.ROUTE_MAX_EXPERTS:
        return None
    return ops.route_plan(expert_indices, num_experts)


def _sort_by_expert(expert_indices: torch.Tensor):
    top_k = expert_indices.shape[1]
    flat_expert = expert_indices.reshape(-1)
    order = torch.argsort(flat_expert, stable=True)
    token_of_slot = torch.div(order, top_k, rounding_mode="floor")
    inverse = torch.empty_like(order)
    inverse[order] = torch.arange(order.numel(), device=order.device)
    return order, token_of_slot, inverse


def dispatch_grouped(x: torch.Tensor, expert_indices: torch.Tensor, num_experts: int):
    """Dispatch for the fused op: rows gathered in expert order plus device-side counts/offsets.

    Returns ``(grouped [T*top_k, K], tokens_per_expert int32 [E], input_offsets int32 [E], inverse)``;
    nothing is read back to the host."""
    plan = _plan_on_device(expert_indices, num_experts)
    if plan is not None:
        tpe, offs, token_of_sorted, pos_of_slot = plan
        return x.index_select(0, token_of_sorted), tpe, offs, pos_of_slot
    _, token_of_slot, inverse = _sort_by_expert(expert_indices)
    grouped = x.index_select(0, token_of_slot)
    tpe = torch.bincount(expert_indices.reshape(-1), minlength=num_experts).to(torch.int32)
    offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
    return grouped, tpe, offs, inverse


def dispatch_indices(expert_indices: torch.Tensor, num_experts: int):
    """Dispatch without moving any activation: ``(row_index int32 [T*top_k], tokens_per_expert,
    input_offsets, inverse)`` -- feed ``row_index`` to ``ops.moe_gather_forward`` and the gather is done
    inside the activation pre-pass."""
    plan = _plan_on_device(expert_indices, num_experts)
    if plan is not None:
        tpe, offs, token_of_sorted, pos_of_slot = plan
        return token_of_sorted, tpe, offs, pos_of_slot
    _, token_of_slot, inverse = _sort_by_expert(expert_indices)
    tpe = torch.bincount(expert_indices.reshape(-1), minlength=num_experts).to(torch.int32)
    offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
    return token_of_slot.to(torch.int32), tpe, offs, inverse


def create_expert_inputs(x: torch.Tensor, routing: RoutingResult, num_experts: int,
                         top_k: int) -> Tuple[List[torch.Tensor], torch.Tensor]:
    """Reference-shaped dispatch: a list of ``[m_e, K]`` tensors and the un-sort permutation."""
    _, token_of_slot, inverse = _sort_by_expert(routing.expert_indices)
    out, o = [], 0
    for e in range(num_experts):
        c = routing.tokens_per_expert[e]
        if c > 0:
            out.append(x[token_of_slot[o:o + c]])
        else:
            out.append(torch.empty(0, x.shape[1], device=x.device, dtype=x.dtype))
        o += c
    return out, inverse


def combine_grouped(grouped_out: torch.Tensor, expert_weights: torch.Tensor, inverse: torch.Tensor,
                    top_k: int) -> torch.Tensor:
    """Un-sort ``[T*top_k, N]`` rows to ``[T, top_k, N]`` and take the routing-weighted sum."""
    if grouped_out.is_cuda and grouped_out.dtype == torch.float32 and 0 < expert_weights.shape[0] <= 65535:
        from . import ops
        return ops.combine(grouped_out, inverse, expert_weights)         # one launch
    y = grouped_out.index_select(0, inverse)
    y = y.view(y.shape[0] // top_k, top_k, y.shape[-1])
    return (y * expert_weights.unsqueeze(-1).to(y.dtype)).sum(dim=1)


def combine_expert_outputs(expert_outputs: List[torch.Tensor], routing: RoutingResult,
                           permutation: torch.Tensor, top_k: int) -> torch.Tensor:
    """Reference-shaped combine (routing.py:152-189)."""
    return combine_grouped(torch.cat(expert_outputs, dim=0), routing.expert_weights, permutation, top_k)


def get_expert_sizes_for_benchmark(num_tokens: int, num_experts: int, hidden_dim: int, ffn_dim: int,
                                   distribution: str = "skewed", device: str = "cuda"):
    """(m_sizes, K, N) of the per-expert GEMMs; ``top_k`` fixed at 2 as in routing.py:217."""
    r = simulate_routing(num_tokens, num_experts, 2, distribution, device)
    return r.tokens_per_expert, hidden_dim, ffn_dim
