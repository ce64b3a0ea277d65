"""Synthetic GPT-2-shaped block stack used for measurement: random weights,
hidden states in / hidden states out.  Per layer:
    LN -> FlashSelfAttention (fused qkv GEMM, tiled attention, out-proj GEMM + residual)
       -> LN -> FusedTransformerMLP (fc1 GEMM + act, fc2 GEMM + residual)
All compute runs in the HIP kernels; residual adds are fused into the GEMM epilogues."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops
from .kernels.attention.flash_attention import FlashAttentionConfig, FlashSelfAttention
from .kernels.attention.ring_attention import RingAttentionConfig, RingCrossAttention
from .kernels.mlp.fused_mlp import FusedMLPConfig, FusedTransformerMLP
from ._nn import ResidualStream


class FusedLayerNorm(nn.LayerNorm):
    """nn.LayerNorm whose forward is the HIP row kernel (reference layernorm_kernels.py:191-276)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise ValueError("HIP kernels require input tensors to be on a CUDA (ROCm) device.")
        w = self.weight if self.weight.dtype == x.dtype else self.weight.to(x.dtype)  # fp32 parameters, 16-bit activations
        b = self.bias if (self.bias is None or self.bias.dtype == x.dtype) else self.bias.to(x.dtype)
        return ops.layernorm(x, w, b, self.eps)


class Block(nn.Module):
    def __init__(self, d: int, H: int, I: int, causal: bool, precision: str, activation: str = "gelu"):
        super().__init__()
        acfg = FlashAttentionConfig(causal=causal, precision=precision)
        self.ln_1 = FusedLayerNorm(d)
        self.attn = FlashSelfAttention(d, H, acfg)
        self.ln_2 = FusedLayerNorm(d)
        self.mlp = FusedTransformerMLP(d, I, activation, FusedMLPConfig(precision=precision))

    def stream_ok(self, B: int, S: int, dtype: torch.dtype) -> bool:
        """The block can run with both LayerNorms folded into the GEMMs around them (mio._nn.ResidualStream)."""
        a_ok, m_ok = getattr(self.attn, "stream_ok", None), getattr(self.mlp, "stream_ok", None)  # converted sub-layers (tensor /
        if a_ok is None or m_ok is None:                                                           # sequence parallel) have none
            return False
        return a_ok(B, S, dtype, self.ln_1) and m_ok(B, S, dtype, self.ln_2)

    def forward(self, x, stream_out: bool = False, fold: bool = True):
        """x: [B, S, d] tensor or the ResidualStream of the previous block (fold=False: every LayerNorm as its own kernel).  Where stream_ok(): ln_2 is folded into the output
        projection (statistics) and fc1 (normalisation), and with stream_out the next block's ln_1 into fc2 and its QKV GEMM --
        the residual stream then never leaves the blocked layout and no LayerNorm kernel runs between the GEMMs."""
        B, S, _ = x.shape
        if isinstance(x, ResidualStream) or (fold and self.stream_ok(B, S, x.dtype)):
            a = self.attn(x, residual=x, pre_norm=self.ln_1, stream_out=True)
            return self.mlp(a, residual=a, pre_norm=self.ln_2, stream_out=stream_out)
        if stream_out:
            raise ValueError("stream_out needs a size / dtype with stream_ok()")
        # pre_norm=: the module applies the LayerNorm itself (same kernels; at large sizes LayerNorm then writes its
        # output in the blocked layout the following GEMM fetches contiguously)
        x = self.attn(x, residual=x, pre_norm=self.ln_1)
        return self.mlp(x, residual=x, pre_norm=self.ln_2)


class CrossBlock(nn.Module):
    """Diffusion-style block of BASELINE config 5: LN -> non-causal cross attention (q from x, k / v from a separate
    context tensor; the reference's RingCrossAttention projections, ring_attention.py:413-669) + residual -> LN ->
    FusedMLP-GELU + residual."""

    def __init__(self, d: int, H: int, I: int, precision: str = "bf16", activation: str = "gelu"):
        super().__init__()
        self.ln_1 = FusedLayerNorm(d)
        self.attn = RingCrossAttention(d, H, RingAttentionConfig(precision=precision))
        self.ln_2 = FusedLayerNorm(d)
        self.mlp = FusedTransformerMLP(d, I, activation, FusedMLPConfig(precision=precision))

    def stream_ok(self, B: int, S: int, dtype: torch.dtype) -> bool:
        a_ok, m_ok = getattr(self.attn, "stream_ok", None), getattr(self.mlp, "stream_ok", None)
        if a_ok is None or m_ok is None:
            return False
        return a_ok(B, S, dtype, self.ln_1) and m_ok(B, S, dtype, self.ln_2)

    def forward(self, x, context: torch.Tensor, stream_out: bool = False, fold: bool = True):
        """x: [B, S, d] tensor or the previous block's ResidualStream; the LayerNorms are folded into the GEMMs around them where
        stream_ok() (Block.forward)."""
        B, S, _ = x.shape
        if isinstance(x, ResidualStream) or (fold and self.stream_ok(B, S, x.dtype)):
            a = self.attn(x, context, residual=x, pre_norm=self.ln_1, stream_out=True)
            return self.mlp(a, residual=a, pre_norm=self.ln_2, stream_out=stream_out)
        if stream_out:
            raise ValueError("stream_out needs a size / dtype with stream_ok()")
        x = self.attn(self.ln_1(x), context, residual=x)
        return self.mlp(x, residual=x, pre_norm=self.ln_2)


class CrossAttentionStack(nn.Module):
    """L CrossBlocks over one shared context (d=1280, H=16 -> Dh 80, I=5120 is BASELINE config 5)."""

    def __init__(self, hidden_size: int = 1280, num_heads: int = 16, num_layers: int = 4,
                 intermediate_size: Optional[int] = None, precision: str = "bf16", seed: int = 0):
        super().__init__()
        I = intermediate_size or 4 * hidden_size
        self.h = nn.ModuleList([CrossBlock(hidden_size, num_heads, I, precision) for _ in range(num_layers)])
        self.no_ln_fold = False  # True: every LayerNorm runs as its own kernel (A/B of the fold)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.02)
                    m.bias.zero_()

    @torch.no_grad()
    def forward(self, x: torch.Tensor, context: torch.Tensor) -> torch.Tensor:
        B, S, _ = x.shape
        fold = not self.no_ln_fold and all(blk.stream_ok(B, S, x.dtype) for blk in self.h)
        last = len(self.h) - 1
        for i, blk in enumerate(self.h):
            x = blk(x, context, stream_out=(fold and i < last), fold=fold)
        return x


class GPT2ShapedStack(nn.Module):
    """d=1024, H=16, L=24 is the GPT-2-medium shape of BASELINE config 2."""

    def __init__(self, hidden_size: int = 1024, num_heads: int = 16, num_layers: int = 24,
                 intermediate_size: Optional[int] = None, causal: bool = True, precision: str = "bf16",
                 activation: str = "gelu", seed: int = 0):
        super().__init__()
        I = intermediate_size or 4 * hidden_size
        self.h = nn.ModuleList([Block(hidden_size, num_heads, I, causal, precision, activation)
                                for _ in range(num_layers)])
        self.ln_f = FusedLayerNorm(hidden_size)
        self.no_ln_fold = False  # True: every LayerNorm runs as its own kernel (A/B of the fold, tools/ln_fold_ab.py)
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():  # N(0, 0.02) weights, zero biases (flash_attention.py:534-542)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.02)
                    m.bias.zero_()

    @torch.no_grad()
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, S, _ = x.shape
        fold = not self.no_ln_fold and all(blk.stream_ok(B, S, x.dtype) for blk in self.h)
        last = len(self.h) - 1
        for i, blk in enumerate(self.h):
            x = blk(x, stream_out=(fold and i < last), fold=fold)  # the last block hands a plain tensor to ln_f
        return self.ln_f(x)
