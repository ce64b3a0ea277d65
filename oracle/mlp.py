"""CPU oracle: FusedMLP (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py)."""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F


def gelu_tanh(x: torch.Tensor) -> torch.Tensor:
    """0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3))) -- kernels/mlp/fused_mlp.py:227-231 and
    the Triton kernel's activation, kernels/triton/mlp_kernels.py:144-161."""
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x.pow(3))))


ACTIVATIONS = ("gelu", "gelu_erf", "relu", "silu", "swiglu")


def fused_mlp(
    hidden_states: torch.Tensor,
    fc1_weight: torch.Tensor,
    fc1_bias: Optional[torch.Tensor],
    fc2_weight: torch.Tensor,
    fc2_bias: Optional[torch.Tensor],
    activation: str = "gelu",
    fc1_gate_weight: Optional[torch.Tensor] = None,
    fc1_gate_bias: Optional[torch.Tensor] = None,
    residual: Optional[torch.Tensor] = None,
    dtype=torch.float64,
) -> torch.Tensor:
    """fc2(act(fc1(x))) with every operand up-cast to `dtype`.

    activation:
      "gelu"      tanh-GELU: FusedTransformerMLP("gelu") -> FusedMLPGeluTanh._forward_pytorch
                  (kernels/mlp/fused_mlp.py:223-237) == the Triton kernel (mlp_kernels.py:144-161)
      "gelu_erf"  exact GELU: FusedMLP base class with activation_fn="gelu" (fused_mlp.py:162-163)
                  == pytorch_fused_mlp("gelu") (mlp_kernels.py:782-783)
      "relu"      fused_mlp.py:164-165 / mlp_kernels.py:784-785
      "silu"      fused_mlp.py:166-167
      "swiglu"    silu(fc1_gate(x)) * fc1(x): fused_mlp.py:262-275 / mlp_kernels.py:786-797
    `residual` (not in the reference functional form) is added after fc2 -- the fused epilogue the
    synthetic block stack uses; residual=None is the reference function.
    """
    x = hidden_states.to(dtype)
    w1, w2 = fc1_weight.to(dtype), fc2_weight.to(dtype)
    b1 = None if fc1_bias is None else fc1_bias.to(dtype)
    b2 = None if fc2_bias is None else fc2_bias.to(dtype)
    h = F.linear(x, w1, b1)
    if activation == "gelu":
        h = gelu_tanh(h)
    elif activation == "gelu_erf":
        h = F.gelu(h)
    elif activation == "relu":
        h = F.relu(h)
    elif activation == "silu":
        h = F.silu(h)
    elif activation == "swiglu":
        if fc1_gate_weight is None:
            raise ValueError("SwiGLU activation requires gate weights")
        g = F.linear(x, fc1_gate_weight.to(dtype), None if fc1_gate_bias is None else fc1_gate_bias.to(dtype))
        h = F.silu(g) * h
    else:
        raise ValueError(f"Unsupported activation function: {activation}")
    y = F.linear(h, w2, b2)
    if residual is not None:
        y = y + residual.to(dtype)
    return y
