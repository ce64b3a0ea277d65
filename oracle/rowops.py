"""CPU oracle: row-wise ops either side of the hot path (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

from typing import Optional

import torch


def layernorm(x, weight, bias=None, eps: float = 1e-5, dtype=torch.float64):
    """Restates pytorch_layernorm (kernels/triton/layernorm_kernels.py:279-311):
    u = mean, s = mean((x-u)^2) (biased), (x-u)/sqrt(s+eps)*weight + bias."""
    xf = x.to(dtype)
    u = xf.mean(dim=-1, keepdim=True)
    s = (xf - u).pow(2).mean(dim=-1, keepdim=True)
    y = (xf - u) / torch.sqrt(s + eps)
    return weight.to(dtype) * y + (bias.to(dtype) if bias is not None else 0.0)


def layernorm_residual(x, residual, weight, bias=None, eps: float = 1e-5, residual_alpha: float = 1.0,
                       dtype=torch.float64):
    """x + alpha*residual first (layernorm_kernels.py:299-301), then LayerNorm. Returns
    (normalised, summed) -- the sum is what the next residual branch consumes."""
    summed = x.to(dtype) + residual_alpha * residual.to(dtype)
    return layernorm(summed, weight, bias, eps, dtype), summed
