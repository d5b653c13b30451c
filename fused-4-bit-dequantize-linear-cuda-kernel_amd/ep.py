"""Expert-parallel MoE: experts sharded over the ranks of one node, token dispatch / combine by
all-to-all (RCCL over xGMI on MI355X; ``torch.distributed`` backend ``"nccl"`` is RCCL on ROCm).

The reference is single-GPU; the dispatch / combine *semantics* are those of its
benchmark/moe_grouped_gemm/routing.py:96-189 (sort rows by expert, gather, per-expert GEMM, un-sort,
routing-weighted sum).  The distribution is new:

  * rank r owns experts [r*E/G, (r+1)*E/G) (contiguous block) -- only their packed weights live on r;
  * every rank owns a slice of the tokens (data-parallel token ownership);
  * step: (1) all-to-all of the per-expert row counts, (2) all-to-all of the rows, uneven splits,
    (3) ONE local grouped INT4 GEMM launch over the received rows, (4) all-to-all back,
    (5) un-sort and routing-weighted sum on the token's home rank.

MI355X has 7 xGMI links per GPU in a full mesh, so each peer pair of an all-to-all has its own
link and the exchange is not ring-bound; at batch 512 the messages are ~0.3-0.7 MB per peer and the
step is latency-dominated.

Because the integer dot products of the GEMM are exact, a row's result does not depend on which
rank computed it: the G-way output equals the 1-GPU grouped output bit for bit.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch
import torch.distributed as dist


def _local_grouped_gemm(packed, scales, zps, precision):
    from . import ops

    def fn(rows, tokens_per_expert, input_offsets):
        return ops.moe_forward(packed, scales, zps, rows, None, tokens_per_expert, input_offsets,
                               precision=precision)

    def gather_fn(rows, row_index, tokens_per_expert, input_offsets, row_weight=None):
        """Same, grouped row i = rows[row_index[i]]: the regrouping copy is fused into the activation pre-pass;
        ``row_weight``: the routing weight of every grouped row, applied in the GEMM epilogue."""
        return ops.moe_gather_forward(packed, scales, zps, rows, row_index, tokens_per_expert, input_offsets,
                                      precision=precision, row_weight=row_weight)
    fn.gather = gather_fn
    return fn


class ExpertParallelMoE:
    """Holds this rank's expert shard and runs dispatch -> grouped GEMM -> combine.

    ``expert_fn(rows [R,K], tokens_per_expert int32 [E_local], input_offsets int32 [E_local]) -> [R,N]``
    defaults to the fused HIP grouped GEMM; tests on CPU (gloo) pass their own.
    """

    def __init__(self, num_experts: int, packed_local: Optional[torch.Tensor] = None,
                 scales_local: Optional[torch.Tensor] = None, zps_local: Optional[torch.Tensor] = None,
                 group=None, precision: str = "default",
                 expert_fn: Optional[Callable] = None, out_features: Optional[int] = None,
                 capacity_factor: Optional[float] = None):
        """``capacity_factor``: None = exact uneven all-to-alls (one small D2H copy of the split sizes per step);
        a number f >= 1 = FIXED-CAPACITY all-to-alls: every rank sends every peer a block of
        ``ceil(f * t_local * top_k / world)`` rows (padded), all sizes are known on the host, the counts stay on the
        device and nothing is read back.  ``f = world`` can never overflow (bit-identical to the exact path);
        smaller factors drop the rows past a peer's capacity (their contribution is zero) and raise the device-side
        flag ``overflowed()`` reports."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        if num_experts % self.world != 0:
            raise ValueError("num_experts must be divisible by the number of ranks")
        self.num_experts = num_experts
        self.experts_per_rank = num_experts // self.world
        if expert_fn is None:
            if packed_local is None or packed_local.shape[0] != self.experts_per_rank:
                raise ValueError("packed_local must hold this rank's experts_per_rank experts")
            expert_fn = _local_grouped_gemm(packed_local, scales_local, zps_local, precision)
            out_features = packed_local.shape[1]
        self.expert_fn = expert_fn
        self.out_features = out_features
        self.last_split = {}
        if capacity_factor is not None and capacity_factor < 1:
            raise ValueError("capacity_factor must be >= 1")
        self.capacity_factor = capacity_factor
        # single-rank device path: routing weights folded into the GEMM epilogue, combine = pure gather-add (SURVEY 8f N1,
        # second half).  Same bits as the weighted combine for top_k <= 2 (tests/test_gpu_w4.py); measured delta in
        # profiles/r03_ab_combine_epilogue.txt.
        self.fold_weights = False
        self._overflow = None           # device-side flag of the fixed-capacity path (accumulates over steps)
        self.record_phases = False      # set True to collect per-phase GPU events of the next device-path steps
        self.phase_events = []          # one list of (name, torch.cuda.Event) per profiled step

    @staticmethod
    def shard(tensor: torch.Tensor, rank: int, world: int) -> torch.Tensor:
        """Rank's contiguous block of a tensor stacked over experts (dim 0)."""
        per = tensor.shape[0] // world
        return tensor[rank * per:(rank + 1) * per]

    def _forward_device(self, x, expert_indices, expert_weights):
        """GPU path with the library's routing kernels: ONE launch plans the dispatch (stable sort by expert, counts,
        offsets, gather and un-sort indices), the gathers are fused into the activation pre-pass, ONE launch
        combines.  Same result as ``forward``'s torch formulation (bit for bit up to top_k = 2, where the
        weighted sum has a single addition)."""
        from . import ops
        G, EL = self.world, self.experts_per_rank
        top_k = expert_indices.shape[1]
        dev = x.device
        K = x.shape[1]
        marks = [] if self.record_phases else None

        def mark(name):
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                marks.append((name, ev))
        mark("start")
        counts, offsets, token_of_sorted, pos_of_slot = ops.route_plan(expert_indices, self.num_experts)
        fused_gather = hasattr(self.expert_fn, "gather") and K % 32 == 0 and x.dtype == torch.float32
        folded = False
        if G == 1:
            if fused_gather and self.fold_weights and x.shape[0] <= 65535:
                w_sorted = torch.empty(pos_of_slot.numel(), dtype=torch.float32, device=dev)
                w_sorted[pos_of_slot.long()] = expert_weights.reshape(-1).to(torch.float32)
                y_sorted = self.expert_fn.gather(x, token_of_sorted, counts, offsets, row_weight=w_sorted)
                folded = True
            elif fused_gather:
                y_sorted = self.expert_fn.gather(x, token_of_sorted, counts, offsets)
            else:
                y_sorted = self.expert_fn(x.index_select(0, token_of_sorted.long()), counts, offsets)
        else:
            send_rows = x.index_select(0, token_of_sorted.long())
            send_counts = counts.to(torch.int64)
            recv_counts = torch.empty(G * EL, dtype=torch.int64, device=dev)
            mark("plan_and_pack")
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
            recv_counts = recv_counts.view(G, EL)
            # split sizes must be host integers for all_to_all_single: one small D2H copy per step
            sizes = torch.stack([send_counts.view(G, EL).sum(1), recv_counts.sum(1)]).cpu()
            in_splits, out_splits = sizes[0].tolist(), sizes[1].tolist()
            R = sum(out_splits)
            recv_rows = torch.empty((R, K), dtype=x.dtype, device=dev)
            mark("counts_exchange_and_host_sizes")
            dist.all_to_all_single(recv_rows, send_rows, out_splits, in_splits, group=self.group)
            mark("dispatch_all_to_all")
            # received order is (source rank, local expert); the GEMM wants (local expert, source rank)
            tpe, offs, gather, scatter = ops.regroup_index(recv_counts, R)
            if R == 0:
                y_recv_order = torch.empty((0, self.out_features), dtype=torch.float32, device=dev)
            elif fused_gather:
                y_recv_order = self.expert_fn.gather(recv_rows, gather, tpe, offs).index_select(0, scatter.long())
            else:
                y_recv_order = self.expert_fn(recv_rows.index_select(0, gather.long()), tpe, offs).index_select(
                    0, scatter.long())
            y_sorted = torch.empty((send_rows.shape[0], y_recv_order.shape[1]), dtype=y_recv_order.dtype, device=dev)
            mark("regroup_and_grouped_gemm")
            dist.all_to_all_single(y_sorted, y_recv_order.contiguous(), in_splits, out_splits, group=self.group)
            mark("combine_all_to_all")
            self.last_split = {"dispatch_rows_sent": in_splits, "dispatch_rows_received": out_splits}
        if G == 1:
            mark("plan_and_grouped_gemm")
        if folded:
            out = ops.combine(y_sorted, pos_of_slot, None, top_k=top_k)
        elif y_sorted.dtype == torch.float32 and x.shape[0] <= 65535:
            out = ops.combine(y_sorted, pos_of_slot, expert_weights)
        else:
            y = y_sorted.index_select(0, pos_of_slot.long()).view(x.shape[0], top_k, -1)
            out = (y * expert_weights.unsqueeze(-1).to(y.dtype)).sum(dim=1)
        mark("weighted_combine")
        if marks is not None:
            self.phase_events.append(marks)
        return out

    def phase_times_ms(self):
        """Mean GPU time of each phase over the profiled steps (call after torch.cuda.synchronize())."""
        acc, n = {}, 0
        for marks in self.phase_events:
            for (_, e0), (name, e1) in zip(marks[:-1], marks[1:]):
                acc[name] = acc.get(name, 0.0) + e0.elapsed_time(e1)
            n += 1
        return {k: v / n for k, v in acc.items()} if n else {}

    def overflowed(self) -> bool:
        """Fixed-capacity path: did any step so far drop rows (one D2H read, on demand only)?"""
        return bool(self._overflow.item()) if self._overflow is not None else False

    def _forward_fixed_capacity(self, x, expert_indices, expert_weights):
        """Dispatch / combine with equal-split all-to-alls (see ``capacity_factor``): no host read-back in the step.
        Rows keep the order of the exact path inside every (source rank, expert) run, the local grouped GEMM sees the
        same rows at the same offsets, so with a capacity that cannot overflow the result is the exact path's, bit for
        bit.  The grouped call gets a row buffer of the worst-case size (world x capacity) and DEVICE-side counts: rows
        past the counts are never computed."""
        G, EL = self.world, self.experts_per_rank
        t, top_k = expert_indices.shape
        dev = x.device
        K = x.shape[1]
        S = t * top_k
        C = min(S, int(-(-self.capacity_factor * S // G)))                  # rows per peer, a host constant
        flat_expert = expert_indices.reshape(-1)
        order = torch.argsort(flat_expert, stable=True)
        sorted_expert = flat_expert.index_select(0, order)
        token_of_slot = torch.div(order, top_k, rounding_mode="floor")
        counts = torch.bincount(flat_expert, minlength=self.num_experts)    # [E] int64, device
        exp_off = torch.cumsum(counts, 0) - counts                          # first sorted slot of every expert
        rank_off = exp_off.view(G, EL)[:, 0]                                # ... of every destination rank
        dest_rank = torch.div(sorted_expert, EL, rounding_mode="floor")
        pos = torch.arange(S, device=dev) - rank_off.index_select(0, dest_rank)   # position inside the peer's block
        valid = pos < C
        over = (~valid).any()
        self._overflow = over if self._overflow is None else (self._overflow | over)
        slot = torch.where(valid, dest_rank * C + pos, torch.full_like(pos, G * C))   # dropped rows -> a dummy row
        send_buf = torch.zeros((G * C + 1, K), dtype=x.dtype, device=dev)
        send_buf[slot] = x.index_select(0, token_of_slot)
        # counts after clipping every peer's block at C rows (rows are in expert order inside a block)
        in_block = (exp_off.view(G, EL) - rank_off[:, None])
        sent = (torch.clamp(in_block + counts.view(G, EL), max=C) - torch.clamp(in_block, max=C)).reshape(-1)
        recv_counts = torch.empty(G * EL, dtype=torch.int64, device=dev)
        dist.all_to_all_single(recv_counts, sent.contiguous(), group=self.group)
        recv_buf = torch.empty((G * C, K), dtype=x.dtype, device=dev)
        dist.all_to_all_single(recv_buf, send_buf[:G * C].contiguous(), group=self.group)
        # received layout: [source rank][C rows, its local experts in order]; the GEMM wants (local expert, source rank)
        cnt = recv_counts.view(G, EL)
        src_incl = torch.cumsum(cnt, 1)                                     # [G, EL]
        p = torch.arange(C, device=dev).expand(G, C).contiguous()
        e_of = torch.searchsorted(src_incl, p, right=True)                  # local expert of received slot (s, p); EL = padding
        is_row = e_of < EL
        e_cl = torch.clamp(e_of, max=EL - 1)
        within = p - (src_incl - cnt).gather(1, e_cl)
        exp_major_cnt = cnt.t().contiguous()                                # [EL, G]
        exp_major_off = (torch.cumsum(exp_major_cnt.reshape(-1), 0) - exp_major_cnt.reshape(-1)).view(EL, G)
        s_idx = torch.arange(G, device=dev)[:, None].expand(G, C)
        dest = torch.where(is_row, exp_major_off[e_cl, s_idx] + within, torch.full_like(p, G * C)).reshape(-1)
        grouped = torch.zeros((G * C + 1, K), dtype=x.dtype, device=dev)
        grouped[dest] = recv_buf
        tpe = cnt.sum(0).to(torch.int32)
        offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
        y_grouped = self.expert_fn(grouped[:G * C], tpe, offs)              # [G * C, N]; rows past the counts are zero
        y_pad = torch.cat([y_grouped, y_grouped.new_zeros((1, y_grouped.shape[1]))])
        y_recv_order = y_pad.index_select(0, dest)
        y_back = torch.empty_like(y_recv_order)
        dist.all_to_all_single(y_back, y_recv_order.contiguous(), group=self.group)
        y_back = torch.cat([y_back, y_back.new_zeros((1, y_back.shape[1]))])
        y_sorted = y_back.index_select(0, slot)                             # dropped rows read the zero row
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(S, device=dev)
        y = y_sorted.index_select(0, inverse).view(t, top_k, -1)
        return (y * expert_weights.unsqueeze(-1).to(y.dtype)).sum(dim=1)

    def forward(self, x: torch.Tensor, expert_indices: torch.Tensor, expert_weights: torch.Tensor) -> torch.Tensor:
        """x [t_local, K] float32, expert_indices / expert_weights [t_local, top_k] -> [t_local, N]."""
        if self.capacity_factor is not None and self.world > 1 and x.shape[0] > 0:
            return self._forward_fixed_capacity(x, expert_indices, expert_weights)
        if x.is_cuda and self.num_experts <= 128 and x.shape[0] > 0:
            return self._forward_device(x, expert_indices, expert_weights)
        G, EL = self.world, self.experts_per_rank
        top_k = expert_indices.shape[1]
        dev = x.device
        # ---- sort this rank's (token, slot) pairs by global expert id == by destination rank
        flat_expert = expert_indices.reshape(-1)
        order = torch.argsort(flat_expert, stable=True)
        token_of_slot = torch.div(order, top_k, rounding_mode="floor")
        send_rows = x.index_select(0, token_of_slot)
        send_counts = torch.bincount(flat_expert, minlength=self.num_experts).to(torch.int64)   # [E]

        if G == 1:
            tpe = send_counts.to(torch.int32)
            offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
            y_sorted = self.expert_fn(send_rows, tpe, offs)
        else:
            # ---- (1) counts: recv_counts[s, e] = rows rank s sends for my local expert e
            recv_counts = torch.empty(G * EL, dtype=torch.int64, device=dev)
            dist.all_to_all_single(recv_counts, send_counts, group=self.group)
            recv_counts = recv_counts.view(G, EL)
            # split sizes must be host integers for all_to_all_single: one small D2H copy per step
            sizes = torch.stack([send_counts.view(G, EL).sum(1), recv_counts.sum(1)]).cpu()
            in_splits, out_splits = sizes[0].tolist(), sizes[1].tolist()
            # ---- (2) dispatch rows
            recv_rows = torch.empty((sum(out_splits), x.shape[1]), dtype=x.dtype, device=dev)
            dist.all_to_all_single(recv_rows, send_rows, out_splits, in_splits, group=self.group)
            # received order is (source rank, local expert); the GEMM wants (local expert, source rank)
            cnt = recv_counts                                             # [G, EL]
            src_major_off = (torch.cumsum(cnt.reshape(-1), 0) - cnt.reshape(-1)).view(G, EL)
            exp_major_cnt = cnt.t().contiguous()                          # [EL, G]
            exp_major_off = (torch.cumsum(exp_major_cnt.reshape(-1), 0) - exp_major_cnt.reshape(-1)).view(EL, G)
            R = recv_rows.shape[0]
            # position in expert-major order of every received row
            seg_of_row = torch.repeat_interleave(torch.arange(G * EL, device=dev), cnt.reshape(-1), output_size=R)
            s_idx, e_idx = torch.div(seg_of_row, EL, rounding_mode="floor"), seg_of_row % EL
            within = torch.arange(R, device=dev) - src_major_off.reshape(-1)[seg_of_row]
            dest = exp_major_off[e_idx, s_idx] + within
            grouped = torch.empty_like(recv_rows)
            grouped[dest] = recv_rows
            tpe = cnt.sum(0).to(torch.int32)
            offs = (torch.cumsum(tpe, 0) - tpe).to(torch.int32)
            # ---- (3) one grouped launch over all local experts
            y_grouped = self.expert_fn(grouped, tpe, offs)
            y_recv_order = y_grouped[dest]
            # ---- (4) combine: send results back along the reverse routes
            y_sorted = torch.empty((send_rows.shape[0], y_recv_order.shape[1]), dtype=y_recv_order.dtype, device=dev)
            dist.all_to_all_single(y_sorted, y_recv_order.contiguous(), in_splits, out_splits, group=self.group)
            self.last_split = {"dispatch_rows_sent": in_splits, "dispatch_rows_received": out_splits}
        # ---- (5) un-sort to [t_local, top_k, N] and take the routing-weighted sum
        inverse = torch.empty_like(order)
        inverse[order] = torch.arange(order.numel(), device=dev)
        y = y_sorted.index_select(0, inverse).view(x.shape[0], top_k, -1)
        return (y * expert_weights.unsqueeze(-1).to(y.dtype)).sum(dim=1)

    __call__ = forward
