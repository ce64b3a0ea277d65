"""Ring-attention modules behind the reference's surface (kernels/attention/ring_attention.py).

Semantics ("A"): exact softmax attention over KV chunks with a running (max, sum, acc) state --
the reference's Triton kernel `_ring_attention_forward_kernel` and its PyTorch fallback
(kernels/triton/attention_kernels.py:35-202, 1520-1591), and `RingCrossAttention._ring_cross_attention`
(ring_attention.py:597-660).  On one GPU that is one launch of the tiled HIP kernel (the KV loop IS the chunk
loop); across GPUs the same kernel runs once per ring step with the (o_acc, lse) carry
(mio.parallelism.sequence_parallel.ring_attention).  `RingSelfAttention._ring_self_attention`
(:349-372) is a different, non-exact recurrence in the reference and is deliberately not reproduced.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn

from ... import ops
from ..._nn import CastCache, ResidualStream, compute_dtype, linear


@dataclass
class RingAttentionConfig:
    """Mirror of RingAttentionConfig (ring_attention.py:40-89); same fields, same validation errors.
    `use_triton` is accepted for signature compatibility: the HIP kernel is the only compute path."""
    world_size: int = 1
    chunk_size: Optional[int] = None
    fuse_qkv: bool = True
    use_flash_attention: bool = False
    use_triton: bool = True
    precision: str = "bf16"
    communication_dtype: torch.dtype = torch.bfloat16
    normalize_attention_scores: bool = True
    attention_dropout: float = 0.0

    def __post_init__(self):
        if self.world_size < 1:
            raise ValueError(f"world_size must be >= 1, got {self.world_size}")
        if self.chunk_size is not None and self.chunk_size <= 0:
            raise ValueError(f"chunk_size must be > 0 if specified, got {self.chunk_size}")
        if self.precision not in ["fp32", "fp16", "bf16"]:
            raise ValueError(f"precision must be one of ['fp32', 'fp16', 'bf16'], got {self.precision}")
        if self.attention_dropout < 0 or self.attention_dropout >= 1:
            raise ValueError(f"attention_dropout must be in [0, 1), got {self.attention_dropout}")
        self.compute_dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[self.precision]


class RingAttention(nn.Module):
    """Base class (ring_attention.py:92-166)."""

    def __init__(self, hidden_size: int, num_attention_heads: int, config: RingAttentionConfig):
        super().__init__()
        self.hidden_size = hidden_size
        self.num_attention_heads = num_attention_heads
        self.config = config
        self.head_dim = hidden_size // num_attention_heads
        if self.head_dim * num_attention_heads != hidden_size:
            raise ValueError(
                f"hidden_size ({hidden_size}) is not divisible by num_attention_heads ({num_attention_heads})")
        self.scale = 1.0 / math.sqrt(self.head_dim)
        self._cast = CastCache()

    def get_effective_bytes_per_token(self) -> int:
        """Bytes of K+V per token held at a time by one rank (:128-149)."""
        elem = torch.empty((), dtype=self.config.compute_dtype).element_size()
        return 2 * self.hidden_size * elem

    def calculate_theoretical_memory_savings(self, seq_len: int) -> float:
        """Score-matrix bytes a dense implementation would hold / bytes of the tiled form (:151-165)."""
        dense = seq_len * seq_len * self.num_attention_heads
        tiled = seq_len * self.hidden_size * 2
        return dense / max(tiled, 1)

    def _check(self, x: torch.Tensor) -> torch.dtype:
        if x.dim() != 3:
            raise ValueError(f"Expected 3D input tensor, got shape: {x.shape}")
        if not x.is_cuda:
            raise ValueError("HIP kernels require input tensors to be on a CUDA (ROCm) device.")
        if self.training and self.config.attention_dropout > 0.0:
            raise NotImplementedError("attention dropout (training) is not supported by the inference kernel")
        return compute_dtype(self.config.precision, x)

    def _heads(self, t: torch.Tensor) -> torch.Tensor:
        """[B,S,H*D] -> head-major VIEW [B,H,S,D] (the kernel takes strides; nothing is copied)."""
        B, S, _ = t.shape
        return t.view(B, S, self.num_attention_heads, self.head_dim).permute(0, 2, 1, 3)


class RingSelfAttention(RingAttention):
    """Self-attention shell (ring_attention.py:168-410): fused or separate q/k/v projections, `out_proj`."""

    def __init__(self, hidden_size: int, num_attention_heads: int, config: RingAttentionConfig):
        super().__init__(hidden_size, num_attention_heads, config)
        if config.fuse_qkv:
            self.qkv_proj = nn.Linear(hidden_size, 3 * hidden_size, bias=True)
        else:
            self.q_proj = nn.Linear(hidden_size, hidden_size, bias=True)
            self.k_proj = nn.Linear(hidden_size, hidden_size, bias=True)
            self.v_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.attention_dropout = nn.Dropout(config.attention_dropout)

    def prepare_attention_inputs(self, hidden_states: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(q, k, v) as [B,H,S,D] views (:237-273).  The 1/sqrt(D) scale is applied inside the kernel, not to q."""
        dt = self._check(hidden_states)
        x = hidden_states if hidden_states.dtype == dt else hidden_states.to(dt)
        B, S, d = x.shape
        c = self._cast
        if self.config.fuse_qkv:
            qkv = linear(x, self.qkv_proj, c, dt).view(B, S, 3, self.num_attention_heads, self.head_dim)
            q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        else:
            q = self._heads(linear(x, self.q_proj, c, dt))
            k = self._heads(linear(x, self.k_proj, c, dt))
            v = self._heads(linear(x, self.v_proj, c, dt))
        return q, k, v

    def forward(self, hidden_states: torch.Tensor, attention_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """hidden_states [B,S,d]; attention_mask additive, broadcastable to [B,1|H,S,S] (:200-235)."""
        in_dtype = hidden_states.dtype
        q, k, v = self.prepare_attention_inputs(hidden_states)
        dt = q.dtype
        if attention_mask is not None and attention_mask.dim() == 4 and attention_mask.shape[2] == 1:
            attention_mask = attention_mask.expand(-1, -1, q.shape[2], -1)  # [B,1,1,S] key mask -> per query row
        ctx = ops.ring_attention_forward(q, k, v, attention_mask)
        out = linear(ctx, self.out_proj, self._cast, dt)
        return out if out.dtype == in_dtype else out.to(in_dtype)


class RingCrossAttention(RingAttention):
    """Cross-attention shell (ring_attention.py:413-669): q from `query_states`, k/v from `key_value_states`."""

    def __init__(self, hidden_size: int, num_attention_heads: int, config: RingAttentionConfig):
        super().__init__(hidden_size, num_attention_heads, config)
        self.q_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.k_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.v_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.out_proj = nn.Linear(hidden_size, hidden_size, bias=True)
        self.attention_dropout = nn.Dropout(config.attention_dropout)

    def stream_ok(self, B: int, Sq: int, dtype: torch.dtype, pre_norm: Optional[nn.LayerNorm]) -> bool:
        """True iff forward(...) can take / return the query-side residual stream as a ResidualStream (mio._nn): the folded
        GEMMs on the query projection (LayerNorm in its read-out) and the output projection (blocked stream + row statistics)."""
        d, M = self.hidden_size, B * Sq
        if dtype not in (torch.float16, torch.bfloat16) or pre_norm is None or pre_norm.weight is None or ops.NO_BLOCKED_X:
            return False
        if compute_dtype(self.config.precision, torch.empty(0, dtype=dtype)) != dtype or tuple(pre_norm.normalized_shape) != (d,):
            return False
        return ops.gemm_ln_ok(M, d, d, "none", fold_in=True) and ops.gemm_ln_ok(M, d, d, "none", stats_out=True)

    def forward(self, query_states: torch.Tensor, key_value_states: torch.Tensor,
                attention_mask: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
                pre_norm: Optional[nn.LayerNorm] = None, stream_out: bool = False) -> torch.Tensor:
        """query_states [B,Sq,d], key_value_states [B,Sk,d], additive mask [B,1|H,Sq,Sk] (:442-499).
        residual (not in the reference): added in the out-projection's epilogue.
        pre_norm (not in the reference): a LayerNorm applied to query_states first.  With it, query_states may be a
        ResidualStream and stream_out=True returns one (where stream_ok()): the residual is then the stream itself, the
        LayerNorm runs in the query projection's read-out and the output projection writes the new stream blocked + its row
        statistics (ops.gemm_ln)."""
        if isinstance(query_states, ResidualStream) or stream_out:
            B, Sq, d = query_states.shape
            if (residual is not None and residual is not query_states) or not self.stream_ok(B, Sq, query_states.dtype, pre_norm):
                raise ValueError("the ResidualStream form needs pre_norm, residual = the query input itself and a size with stream_ok()")
            dt, c, M = query_states.dtype, self._cast, B * Sq
            self._check(key_value_states)
            xkv = key_value_states if key_value_states.dtype == dt else key_value_states.to(dt)
            if isinstance(query_states, ResidualStream):
                wfb, bfold = c.get_ln_folded(self.q_proj, pre_norm, dt)
                q2, _ = ops.gemm_ln(query_states.blocked, wfb, bfold, M=M, N=d, K=d, x_blocked=True, ln_stats=query_states.stats,
                                    eps=pre_norm.eps)
                res, res_blocked = query_states.blocked, True
            else:
                xn = ops.layernorm(query_states, c.get(pre_norm.weight, dt), c.get(pre_norm.bias, dt), pre_norm.eps)
                q2 = linear(xn, self.q_proj, c, dt)
                res, res_blocked = query_states.reshape(M, d), False
            q = self._heads(q2.view(B, Sq, d))
            Sk = xkv.shape[1]
            kpre = (attention_mask is None and d % 128 == 0 and d % 32 == 0
                    and ops.fa3_k_prescaled_ok(B, Sq, Sk, self.num_attention_heads, self.head_dim, d, d)
                    and ops.blocked_weight_ok(B * Sk, d, d) and ops.col_scale_ok(B * Sk, d, d))
            cs = (0, d, self.scale * 1.4426950408889634) if kpre else None
            k = self._heads(linear(xkv, self.k_proj, c, dt, col_scale=cs))
            v = self._heads(linear(xkv, self.v_proj, c, dt))
            ctx = ops.ring_attention_forward(q, k, v, attention_mask, k_prescaled=kpre)
            y, st = ops.gemm_ln(ctx.reshape(M, d), c.get_blocked(self.out_proj.weight, dt), c.get(self.out_proj.bias, dt), M=M, N=d,
                                K=d, residual=res, res_blocked=res_blocked, out_blocked=stream_out, stats_out=stream_out)
            return ResidualStream(y, st, (B, Sq, d)) if stream_out else y.view(B, Sq, d)
        if pre_norm is not None:
            query_states = ops.layernorm(query_states, self._cast.get(pre_norm.weight, query_states.dtype),
                                         self._cast.get(pre_norm.bias, query_states.dtype), pre_norm.eps)
        in_dtype = query_states.dtype
        dt = self._check(query_states)
        self._check(key_value_states)
        xq = query_states if query_states.dtype == dt else query_states.to(dt)
        xkv = key_value_states if key_value_states.dtype == dt else key_value_states.to(dt)
        c = self._cast
        q = self._heads(linear(xq, self.q_proj, c, dt))
        # where the kernels allow it the K projection's epilogue hands over K * softmax_scale * log2(e) (one rounding) and
        # the attention kernel drops its per-score multiply (ops.fa3_fwd k_prescaled)
        B, Sq, d = xq.shape
        Sk = xkv.shape[1]
        kpre = (attention_mask is None and d % 128 == 0 and d % 32 == 0
                and ops.fa3_k_prescaled_ok(B, Sq, Sk, self.num_attention_heads, self.head_dim, d, d)
                and ops.blocked_weight_ok(B * Sk, d, d) and ops.col_scale_ok(B * Sk, d, d))
        cs = (0, d, self.scale * 1.4426950408889634) if kpre else None
        k = self._heads(linear(xkv, self.k_proj, c, dt, col_scale=cs))
        v = self._heads(linear(xkv, self.v_proj, c, dt))
        ctx = ops.ring_attention_forward(q, k, v, attention_mask, k_prescaled=kpre)
        r = None if residual is None else (residual if residual.dtype == dt else residual.to(dt))
        out = linear(ctx, self.out_proj, c, dt, residual=r)
        return out if out.dtype == in_dtype else out.to(in_dtype)
