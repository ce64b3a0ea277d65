"""Sequence parallelism / ring attention across the GPUs of one node.

Mirrors reference parallelism/sequence_parallel.py: SequenceParallelConfig (:21-85),
SequenceShardedModule (:88-342), SequenceParallelAttention (:345-640), SequenceParallelMLP (:643-720),
SequenceParallelConverter (:723-920), partition_sequence / gather_sequence (:925-996).

Semantics: "ring" here is EXACT attention -- online softmax carried across the K/V
chunks as an (o, lse) state, semantics (A) of the reference's ring kernel
(kernels/triton/attention_kernels.py:164-193) -- not the per-step-softmax average of
`SequenceParallelAttention._ring_attention` (:555-585), which the reference itself documents as "not
mathematically equivalent to full attention".  "full" all-gathers K/V (:587-640) and is the exactness
cross-check; "local" attends within the shard only (:480-517).

Communication: K/V shards move either around the neighbour ring (sp-1 steps, each overlapped with the
attention of the chunk already present) or over the full xGMI mesh (every peer transfer posted at once;
an MI355X node has a direct link per GPU pair).  Causal attention uses absolute positions
(q_offset/k_offset in the kernel), and optionally zig-zag placement so every rank does equal work.
"""
from __future__ import annotations

import copy
import time
from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import _local
from . import communication as comm


@dataclass
class SequenceParallelConfig:
    """Fields as reference :21-47, plus `exchange` ("ring" | "mesh"), `causal` and `zigzag`."""
    world_size: int = 1
    sp_size: int = 1
    overlap_communication: bool = True
    attention_handling: str = "ring"  # "local", "ring", or "full"
    chunk_size: Optional[int] = None
    buffer_reuse: bool = True
    communication_dtype: torch.dtype = torch.float16
    exchange: str = "mesh"
    causal: bool = False
    zigzag: bool = False
    tp_size: int = 1  # tensor-parallel degree of the same job: sequence groups stride by it (mesh of parallel_utils)

    def __post_init__(self):
        if self.world_size % self.sp_size != 0:
            raise ValueError(f"Sequence parallel size ({self.sp_size}) must divide world size ({self.world_size})")
        if self.attention_handling not in ["local", "ring", "full"]:
            raise ValueError(f"Attention handling strategy '{self.attention_handling}' not supported. "
                             f"Use 'local', 'ring', or 'full'.")
        if self.chunk_size is not None and self.chunk_size <= 0:
            raise ValueError(f"Chunk size must be positive, got {self.chunk_size}")
        if self.exchange not in ("ring", "mesh"):
            raise ValueError("exchange must be 'ring' or 'mesh'")

    def get_sp_group(self) -> Optional[dist.ProcessGroup]:
        if self.sp_size == 1 and not dist.is_initialized():
            return None
        return comm.setup_sequence_parallel_group(self.world_size, self.sp_size, self.tp_size)

    def get_dp_size(self) -> int:
        return self.world_size // (self.sp_size * self.tp_size)

    def get_rank_info(self) -> Tuple[int, int]:
        """(rank inside the sequence-parallel group, data-parallel index) on the mesh rank = (dp * SP + sp) * TP + tp;
        with registered groups (parallel_utils.initialize_parallel_groups) the group itself is asked."""
        rank = comm.get_rank()
        if dist.is_initialized() and self.sp_size > 1:
            grp = self.get_sp_group()
            if grp is not None:
                from . import parallel_utils
                tp = self.tp_size
                if parallel_utils._PARALLEL_GROUPS:
                    tg = parallel_utils._PARALLEL_GROUPS.get("tensor")
                    tp = 1 if tg is None else dist.get_world_size(tg)
                return dist.get_rank(grp), rank // (self.sp_size * tp)
        return (rank // self.tp_size) % self.sp_size, rank // (self.sp_size * self.tp_size)


# ---------------------------------------------------------------------------------------------------
# ring attention core
# ---------------------------------------------------------------------------------------------------
def zigzag_blocks(rank: int, sp: int) -> Tuple[int, int]:
    """Block ids (of 2*sp equal blocks) held by `rank` under zig-zag placement."""
    return rank, 2 * sp - 1 - rank


def zigzag_shard(tensor: torch.Tensor, rank: int, sp: int, seq_dim: int = 1) -> torch.Tensor:
    """This rank's [block r ; block 2sp-1-r] slice of a full sequence."""
    S = tensor.shape[seq_dim]
    if S % (2 * sp) != 0:
        raise ValueError(f"sequence length {S} must be divisible by 2*sp_size ({2 * sp}) for zig-zag placement")
    blk = S // (2 * sp)
    a, b = zigzag_blocks(rank, sp)
    return torch.cat([tensor.narrow(seq_dim, a * blk, blk), tensor.narrow(seq_dim, b * blk, blk)], dim=seq_dim)


def zigzag_unshard(shards: List[torch.Tensor], sp: int, seq_dim: int = 1) -> torch.Tensor:
    """Inverse of zigzag_shard over the list of per-rank shards."""
    blk = shards[0].shape[seq_dim] // 2
    blocks = [None] * (2 * sp)
    for r, s in enumerate(shards):
        a, b = zigzag_blocks(r, sp)
        blocks[a] = s.narrow(seq_dim, 0, blk)
        blocks[b] = s.narrow(seq_dim, blk, blk)
    return torch.cat(blocks, dim=seq_dim)


def ring_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, group: Optional[dist.ProcessGroup] = None, *,
                   layout: str = "bhsd", causal: bool = False, zigzag: bool = False, exchange: str = "mesh",
                   softmax_scale: Optional[float] = None, additive_mask: Optional[torch.Tensor] = None,
                   recv_buffers: Optional[dict] = None, overlap: bool = True, k_prescaled: bool = False) -> torch.Tensor:
    """Exact attention of the local queries over the K/V shards of every rank in `group`.

    q/k/v are this rank's shards ([B,H,S/sp,D] for "bhsd", [B,S/sp,H,D] for "bshd"); returns the local
    output in the same layout.  Per step: one launch of the tiled kernel with the running (o fp32, lse)
    carried in and out; K/V of the next step travel meanwhile.  additive_mask (non-causal only):
    [B,1|H,Sq_local,S_total], columns in global key order.  recv_buffers: a dict kept by the caller between calls --
    the mesh exchange then reuses its receive buffers instead of allocating sp - 1 K/V copies per call.
    overlap=False (measurement only): every transfer completes before the attention that could have hidden it starts.
    k_prescaled: the K shards (local and travelling) already hold K * softmax_scale * log2(e) (ops.fa3_fwd k_prescaled).
    """
    sp = comm.get_world_size(group) if dist.is_initialized() else 1
    r = comm.get_rank(group) if dist.is_initialized() else 0
    si, hi = (2, 1) if layout == "bhsd" else (1, 2)
    B, Sl, H, D = q.shape[0], q.shape[si], q.shape[hi], q.shape[3]
    Skl = k.shape[si]
    dev = q.device
    out = torch.empty_like(q, memory_format=torch.contiguous_format)
    if causal and additive_mask is not None:
        raise ValueError("ring_attention: causal and additive_mask are mutually exclusive")
    if zigzag and not causal:
        raise ValueError("zig-zag placement is only meaningful for causal attention")

    # state: one (o_acc, lse) per query segment (zig-zag has two segments with different positions)
    if zigzag:
        if Sl % 2 or Skl % 2:
            raise ValueError("zig-zag shards must have even length")
        qh, kh = Sl // 2, Skl // 2
        qa, qb = zigzag_blocks(r, sp)
        q_segs = [(0, qh, qa * qh), (qh, qh, qb * qh)]  # (start, len, absolute offset)
    else:
        q_segs = [(0, Sl, r * Sl)]
    states = [(torch.zeros(B, n, H, D, dtype=torch.float32, device=dev),
               torch.full((B, H, n), float("-inf"), dtype=torch.float32, device=dev), False) for (_, n, _) in q_segs]

    def k_segments(origin: int):
        if zigzag:
            ka, kb = zigzag_blocks(origin, sp)
            return [(0, kh, ka * kh), (kh, kh, kb * kh)]
        return [(0, Skl, origin * Skl)]

    # ---- launch plan: visit the local chunk first (its attention runs while the remote K/V travel), then
    # the chunk that originated i ranks upstream, i = 1..sp-1.  Launches whose keys all lie in the future
    # of the query segment are dropped on the host; the LAST launch of each query segment writes `out`.
    origins = [(r - i) % sp for i in range(sp)]
    plan: List[List[Tuple[int, Tuple[int, int, int]]]] = []
    last_launch = {}
    for step, origin in enumerate(origins):
        launches = []
        for qi, (qs, qn, qoff) in enumerate(q_segs):
            for kseg in k_segments(origin):
                if causal and kseg[2] > qoff + qn - 1:
                    continue
                launches.append((qi, kseg))
                last_launch[qi] = (step, len(launches) - 1)
        plan.append(launches)

    def attend(kc: torch.Tensor, vc: torch.Tensor, step: int):
        for j, (qi, (ks, kn, koff)) in enumerate(plan[step]):
            qs, qn, qoff = q_segs[qi]
            o_acc, lse, started = states[qi]
            fin = last_launch[qi] == (step, j)
            kw = dict(layout=layout, causal=causal, softmax_scale=softmax_scale, o_acc=o_acc, lse=lse,
                      carry_in=started, write_out=fin, q_offset=qoff if causal else 0,
                      k_offset=koff if causal else 0)
            if fin:
                kw["out"] = out.narrow(si, qs, qn)
            if additive_mask is not None:
                kw["additive_mask"] = additive_mask[..., koff:koff + kn]
            if k_prescaled:
                kw["k_prescaled"] = True
            _local.attention_step(q.narrow(si, qs, qn), kc.narrow(si, ks, kn), vc.narrow(si, ks, kn), **kw)
            states[qi] = (o_acc, lse, True)

    if sp == 1:
        attend(k, v, 0)
    elif exchange == "mesh":
        handle, chunks = comm.mesh_exchange_start([k, v], group, buffers=recv_buffers)
        if not overlap:
            handle.wait()
            if q.is_cuda:
                torch.cuda.synchronize()
        attend(k, v, 0)
        for i in range(1, sp):
            handle.wait_chunk(i)  # chunk i has landed (RCCL: the one grouped transfer, see mesh_exchange_start)
            attend(chunks[i][0], chunks[i][1], i)
        handle.wait()  # every send has completed before k / v may be freed or overwritten
    else:
        # neighbour ring: step i's K/V arrive from rank-1 while step i-1 is being attended
        k_cur, v_cur = k.contiguous(), v.contiguous()
        for i in range(sp):
            if i < sp - 1:
                h, (k_nxt, v_nxt) = comm.ring_exchange(k_cur, v_cur, group=group, async_op=True)
                if not overlap:
                    h.wait()
                    if q.is_cuda:
                        torch.cuda.synchronize()
            attend(k_cur, v_cur, i)
            if i < sp - 1:
                h.wait()
                k_cur, v_cur = k_nxt, v_nxt
    for qi, (qs, qn, _) in enumerate(q_segs):  # a segment that saw no key at all (cannot happen for self-attention)
        if qi not in last_launch:
            out.narrow(si, qs, qn).zero_()
    return out


# ---------------------------------------------------------------------------------------------------
# module wrappers
# ---------------------------------------------------------------------------------------------------
class SequenceParallelAttention(nn.Module):
    """q/k/v/out Linear + local | ring | full attention over the sequence shard (reference :345-640).
    Input/Output [B, S/sp, hidden]."""

    def __init__(self, hidden_size: int, num_attention_heads: int, config: SequenceParallelConfig,
                 attention_dropout: float = 0.1, head_dim: Optional[int] = None, bias: bool = True):
        super().__init__()
        self.config = config
        self.hidden_size, self.num_attention_heads = hidden_size, num_attention_heads
        self.head_dim = head_dim if head_dim is not None else hidden_size // num_attention_heads
        self.all_head_size = num_attention_heads * self.head_dim
        self.query = nn.Linear(hidden_size, self.all_head_size, bias=bias)
        self.key = nn.Linear(hidden_size, self.all_head_size, bias=bias)
        self.value = nn.Linear(hidden_size, self.all_head_size, bias=bias)
        self.output = nn.Linear(self.all_head_size, hidden_size, bias=bias)
        self.dropout_p = attention_dropout
        self.sp_group = config.get_sp_group() if dist.is_initialized() else None
        self.sp_rank, self.dp_rank = config.get_rank_info()
        for lin in (self.query, self.key, self.value, self.output):  # reference :409-420
            nn.init.xavier_uniform_(lin.weight)
            if lin.bias is not None:
                nn.init.zeros_(lin.bias)
        self.last_communication_time = 0.0
        self._recv_buffers: Optional[dict] = {} if config.buffer_reuse else None  # mesh exchange receive buffers

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                residual: Optional[torch.Tensor] = None, pre_norm: Optional[nn.LayerNorm] = None) -> torch.Tensor:
        if self.training and self.dropout_p > 0:
            raise NotImplementedError("attention dropout (training) is not supported by the inference kernel")
        if pre_norm is not None:  # the pre-LN block's `attn(ln(x))` in one call, as FlashSelfAttention.forward takes it
            hidden_states = _local.layernorm(hidden_states, pre_norm.weight, pre_norm.bias, pre_norm.eps)
        B, Sl, _ = hidden_states.shape
        H, D = self.num_attention_heads, self.head_dim
        cfg = self.config
        mode = cfg.attention_handling
        q = _local.linear(hidden_states, self.query.weight, self.query.bias).view(B, Sl, H, D)
        # ring mode: the K projection scales its columns by softmax_scale * log2(e) in fp32 before their one rounding, and
        # every ring step's attention launch drops its per-score multiply (ops.fa3_fwd k_prescaled) -- where both ends can
        kpre = (mode == "ring" and cfg.sp_size > 1 and attention_mask is None and (H * D) % 128 == 0
                and _local.k_prescale_ok(B, Sl // (2 if cfg.zigzag else 1), H, D, B * Sl, H * D, hidden_states.shape[-1]))
        kcs = (0, H * D, D ** -0.5 * 1.4426950408889634) if kpre else None
        k = _local.linear(hidden_states, self.key.weight, self.key.bias, col_scale=kcs).view(B, Sl, H, D)
        v = _local.linear(hidden_states, self.value.weight, self.value.bias).view(B, Sl, H, D)
        if mode == "local" or cfg.sp_size == 1:
            kw = dict(layout="bshd", causal=cfg.causal)
            if attention_mask is not None:
                kw["additive_mask"] = attention_mask
            ctx = _local.attention_step(q, k, v, **kw)
        elif mode == "ring":
            ctx = ring_attention(q, k, v, self.sp_group, layout="bshd", causal=cfg.causal, zigzag=cfg.zigzag,
                                 exchange=cfg.exchange, additive_mask=attention_mask, recv_buffers=self._recv_buffers,
                                 k_prescaled=kpre)
        else:  # "full": all-gather K/V then one exact attention (reference :587-640)
            t0 = time.perf_counter()
            kf = comm.all_gather(k, dim=1, group=self.sp_group)
            vf = comm.all_gather(v, dim=1, group=self.sp_group)
            self.last_communication_time = time.perf_counter() - t0
            if cfg.zigzag:
                sp = cfg.sp_size
                kf = zigzag_unshard(list(kf.chunk(sp, dim=1)), sp, 1)
                vf = zigzag_unshard(list(vf.chunk(sp, dim=1)), sp, 1)
                outs = []
                for seg, blk in enumerate(zigzag_blocks(self.sp_rank, sp)):
                    h2 = Sl // 2
                    outs.append(_local.attention_step(q[:, seg * h2:(seg + 1) * h2], kf, vf, layout="bshd",
                                                      causal=cfg.causal, q_offset=blk * h2, k_offset=0))
                ctx = torch.cat(outs, dim=1)
            else:
                kw = dict(layout="bshd", causal=cfg.causal, q_offset=self.sp_rank * Sl if cfg.causal else 0, k_offset=0)
                if attention_mask is not None:
                    kw["additive_mask"] = attention_mask
                ctx = _local.attention_step(q, kf, vf, **kw)
        return _local.linear(ctx.reshape(B, Sl, H * D), self.output.weight, self.output.bias, "none", residual)


class SequenceParallelMLP(nn.Module):
    """Token-wise MLP on the local shard: no communication (reference :643-720)."""

    def __init__(self, hidden_size: int, intermediate_size: int, config: SequenceParallelConfig,
                 activation: str = "gelu", bias: bool = True):
        super().__init__()
        self.config = config
        self.dense_h_to_4h = nn.Linear(hidden_size, intermediate_size, bias=bias)
        self.dense_4h_to_h = nn.Linear(intermediate_size, hidden_size, bias=bias)
        self.activation = activation

    def forward(self, hidden_states: torch.Tensor, residual: Optional[torch.Tensor] = None,
                pre_norm: Optional[nn.LayerNorm] = None) -> torch.Tensor:
        if pre_norm is not None:
            hidden_states = _local.layernorm(hidden_states, pre_norm.weight, pre_norm.bias, pre_norm.eps)
        h = _local.linear(hidden_states, self.dense_h_to_4h.weight, self.dense_h_to_4h.bias, self.activation)
        return _local.linear(h, self.dense_4h_to_h.weight, self.dense_4h_to_h.bias, "none", residual)


class SequenceShardedModule(nn.Module):
    """shard [B,S,d] -> wrapped module on [B,S/sp,d] -> gather (reference :88-342)."""

    def __init__(self, module: nn.Module, config: SequenceParallelConfig, seq_dim: int = 1,
                 gather_output: bool = True):
        super().__init__()
        self.module, self.config, self.seq_dim, self.gather_output = module, config, seq_dim, gather_output
        self.sp_group = config.get_sp_group() if dist.is_initialized() else None
        self.last_communication_time = 0.0
        self.last_compute_time = 0.0

    def _shard(self, x: torch.Tensor) -> torch.Tensor:
        sp = self.config.sp_size
        if sp == 1:
            return x
        r = self.config.get_rank_info()[0]
        if self.config.zigzag:
            return zigzag_shard(x, r, sp, self.seq_dim).contiguous()
        return comm.scatter_along_sequence_dim(x, sp, self.sp_group, self.seq_dim)

    def _gather(self, y: torch.Tensor) -> torch.Tensor:
        sp = self.config.sp_size
        if sp == 1 or not self.gather_output:
            return y
        full = comm.all_gather(y, dim=self.seq_dim, group=self.sp_group)
        if self.config.zigzag:
            full = zigzag_unshard(list(full.chunk(sp, dim=self.seq_dim)), sp, self.seq_dim)
        return full

    def forward(self, hidden_states: torch.Tensor, *args, **kwargs):
        x = self._shard(hidden_states)
        t0 = time.perf_counter()
        y = self.module(x, *args, **kwargs)
        self.last_compute_time = time.perf_counter() - t0
        t0 = time.perf_counter()
        out = self._gather(y)
        self.last_communication_time = time.perf_counter() - t0
        return out


class SequenceParallelConverter:
    """Replace attention blocks by SequenceParallelAttention (weights copied) and wrap the model so its
    input is sharded and its output gathered (reference :723-920)."""

    def __init__(self, config: SequenceParallelConfig):
        self.config = config

    def convert_model(self, model: nn.Module, wrap: bool = True) -> nn.Module:
        m = copy.deepcopy(model)
        self._convert(m)
        return SequenceShardedModule(m, self.config) if wrap else m

    def _convert(self, module: nn.Module) -> None:
        from ..kernels.attention.flash_attention import FlashAttentionLayer, FlashSelfAttention

        for name, child in list(module.named_children()):
            if isinstance(child, (FlashAttentionLayer, FlashSelfAttention)):
                d, H = child.hidden_size, child.num_attention_heads
                if child.num_kv_heads != H:
                    raise NotImplementedError("sequence-parallel conversion of GQA attention is not implemented")
                cfg = copy.copy(self.config)
                cfg.causal = child.config.causal
                new = SequenceParallelAttention(d, H, cfg, attention_dropout=0.0)
                p0 = child.o_proj.weight
                new = new.to(device=p0.device, dtype=p0.dtype)
                with torch.no_grad():
                    if isinstance(child, FlashSelfAttention):
                        w, b = child.qkv_proj.weight, child.qkv_proj.bias
                        for i, tgt in enumerate((new.query, new.key, new.value)):
                            tgt.weight.copy_(w[i * d:(i + 1) * d])
                            tgt.bias.copy_(b[i * d:(i + 1) * d])
                    else:
                        for src, tgt in ((child.q_proj, new.query), (child.k_proj, new.key), (child.v_proj, new.value)):
                            tgt.weight.copy_(src.weight)
                            tgt.bias.copy_(src.bias)
                    new.output.weight.copy_(child.o_proj.weight)
                    new.output.bias.copy_(child.o_proj.bias)
                setattr(module, name, new)
            else:
                self._convert(child)


def partition_sequence(tensor: torch.Tensor, sp_size: int, seq_dim: Optional[int] = None) -> List[torch.Tensor]:
    """Reference :925-968: split along the sequence dimension (dim 1 for [B,S,...], dim 0 for [S,...])."""
    if seq_dim is None:
        seq_dim = 1 if tensor.dim() >= 3 else 0
    S = tensor.shape[seq_dim]
    if S % sp_size != 0:
        raise ValueError(f"Sequence length ({S}) must be divisible by sequence parallel size ({sp_size})")
    return list(torch.chunk(tensor, sp_size, dim=seq_dim))


def gather_sequence(tensor_list: List[torch.Tensor], dim: int = 1) -> torch.Tensor:
    """Reference :970-996."""
    return torch.cat(tensor_list, dim=dim)
