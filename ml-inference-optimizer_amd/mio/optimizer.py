"""The reference's plugin surface for this path: Optimizer(model).optimize(use_flash_attention,
use_fused_mlp, tensor_parallel_size) (README.md:57-81 -- the class exists only in the reference's README)
and the two names main.py expects, apply_flash_attention / apply_fused_mlp (main.py:96-104)."""
from __future__ import annotations

from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from .kernels.attention.flash_attention import FlashAttentionConfig, ModelConverter
from .kernels.mlp.fused_mlp import FusedMLPConfig, MLPConverter


def _precision_of(model: nn.Module, default: str = "fp16") -> str:
    p = next(model.parameters(), None)
    if p is not None and p.dtype == torch.bfloat16:
        return "bf16"
    if p is not None and p.dtype == torch.float16:
        return "fp16"
    return default


def apply_flash_attention(model: nn.Module, config: Optional[FlashAttentionConfig] = None) -> nn.Module:
    """Swap attention modules for FlashAttentionLayer / FlashSelfAttention (weights copied)."""
    cfg = config or FlashAttentionConfig(precision=_precision_of(model))
    return ModelConverter(cfg).convert_model(model)


def apply_fused_mlp(model: nn.Module, config: Optional[FusedMLPConfig] = None) -> nn.Module:
    """Swap MLP blocks for FusedTransformerMLP (weights copied)."""
    cfg = config or FusedMLPConfig(precision=_precision_of(model))
    return MLPConverter(cfg).convert_model(model)


def apply_fused_layernorm(model: nn.Module) -> nn.Module:
    """KernelConfig.use_custom_layernorm of the reference (config/config_schema.py:13-19)."""
    from .synthetic import FusedLayerNorm

    for name, child in list(model.named_children()):
        if type(child) is nn.LayerNorm and len(child.normalized_shape) == 1 and child.elementwise_affine:
            new = FusedLayerNorm(child.normalized_shape[0], eps=child.eps).to(device=child.weight.device,
                                                                             dtype=child.weight.dtype)
            new.load_state_dict(child.state_dict())
            setattr(model, name, new)
        else:
            apply_fused_layernorm(child)
    return model


class Optimizer:
    """`Optimizer(model).optimize(...)` returns the (mutated) model; see README.md:57-81 of the reference."""

    def __init__(self, model: nn.Module):
        self.model = model

    def profile(self) -> Dict[str, Any]:
        """Static bottleneck summary: which replaceable blocks the model holds and their parameter share."""
        conv_a, conv_m = ModelConverter(), MLPConverter()
        total = sum(p.numel() for p in self.model.parameters()) or 1
        att = mlp = 0
        n_att = n_mlp = 0
        for m in self.model.modules():
            if conv_a._is_attention_module(m) and conv_a._convertible(m):
                n_att += 1
                att += sum(p.numel() for p in m.parameters())
            elif any(t in m.__class__.__name__ for t in ("MLP", "FFN", "FeedForward")) and conv_m._detect_mlp_type(m):
                n_mlp += 1
                mlp += sum(p.numel() for p in m.parameters())
        return {"attention_modules": n_att, "mlp_modules": n_mlp, "attention_param_fraction": att / total,
                "mlp_param_fraction": mlp / total,
                "bottlenecks": [n for n, c in (("attention", n_att), ("mlp", n_mlp)) if c]}

    def optimize(self, use_flash_attention: bool = True, use_fused_mlp: bool = True, tensor_parallel_size: int = 1,
                 use_custom_layernorm: bool = False, causal: Optional[bool] = None,
                 precision: Optional[str] = None) -> nn.Module:
        model = self.model
        prec = precision or _precision_of(model)
        if use_flash_attention:
            cfg = FlashAttentionConfig(precision=prec, causal=bool(causal))
            model = apply_flash_attention(model, cfg)
        if use_fused_mlp:
            model = apply_fused_mlp(model, FusedMLPConfig(precision=prec))
        if use_custom_layernorm:
            model = apply_fused_layernorm(model)
        if tensor_parallel_size and tensor_parallel_size > 1:
            import torch.distributed as dist
            from .parallelism.tensor_parallel import ModelParallelConverter, TensorParallelConfig

            if not dist.is_initialized():
                raise RuntimeError("tensor_parallel_size > 1 needs torch.distributed to be initialised "
                                   "(mio.parallelism.initialize_distributed)")
            ws = dist.get_world_size()
            model = ModelParallelConverter(TensorParallelConfig(world_size=ws, tp_size=tensor_parallel_size)).convert_model(model)
        self.model = model
        return model
