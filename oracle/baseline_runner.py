"""CPU oracle: the reference's CPU baseline harness (TEST INFRASTRUCTURE ONLY).

Restates `baseline/inference.py`'s working path:
`create_inference_runner(model, "cpu", "fp32", model_type="base")` -> `BasicInferenceRunner`
(:1779-1837); `warmup(inputs, n)` under eval()+no_grad (:616-638); `run_inference` =
`time.perf_counter()` around one no_grad forward, metric `total_time_ms` (:653-713);
tokens/s = batch*seq / avg latency (`benchmarks/runners.py:356-358`).

The model is a plain-PyTorch GPT-2-shaped block stack (random weights from a local shape, never
fetched): per layer LN -> attention (q/k/v/o Linear, dense softmax) -> residual -> LN ->
MLP (Linear, tanh-GELU, Linear) -> residual.  It is also the unconverted model that the
product's converters are tested on.
"""
from __future__ import annotations

import math
import time
from typing import Any, Dict, Tuple

import torch
import torch.nn as nn

from .mlp import gelu_tanh


class PlainSelfAttention(nn.Module):
    """q/k/v/o projection attention with the attribute names ModelConverter sniffs for
    (kernels/attention/flash_attention.py:1048-1059: q_proj/k_proj/v_proj + o_proj + num_heads)."""

    def __init__(self, hidden_size: int, num_heads: int, causal: bool = True):
        super().__init__()
        self.hidden_size, self.num_heads, self.causal = hidden_size, num_heads, causal
        self.head_dim = hidden_size // num_heads
        self.q_proj = nn.Linear(hidden_size, hidden_size)
        self.k_proj = nn.Linear(hidden_size, hidden_size)
        self.v_proj = nn.Linear(hidden_size, hidden_size)
        self.o_proj = nn.Linear(hidden_size, hidden_size)

    def forward(self, hidden_states, attention_mask=None):
        B, S, _ = hidden_states.shape
        H, D = self.num_heads, self.head_dim
        q = self.q_proj(hidden_states).view(B, S, H, D).transpose(1, 2)
        k = self.k_proj(hidden_states).view(B, S, H, D).transpose(1, 2)
        v = self.v_proj(hidden_states).view(B, S, H, D).transpose(1, 2)
        s = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(D)
        if self.causal:
            s = s.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool, device=s.device), 1), -1e9)
        p = torch.softmax(s, dim=-1)
        o = torch.matmul(p, v).transpose(1, 2).reshape(B, S, H * D)
        return self.o_proj(o)


class GELUTanh(nn.Module):
    def forward(self, x):
        return gelu_tanh(x)


class PlainMLP(nn.Module):
    """linear1 -> activation -> linear2: MLPConverter's "pytorch" pattern
    (kernels/mlp/fused_mlp.py:459-482)."""

    def __init__(self, hidden_size: int, intermediate_size: int):
        super().__init__()
        self.linear1 = nn.Linear(hidden_size, intermediate_size)
        self.activation = GELUTanh()
        self.linear2 = nn.Linear(intermediate_size, hidden_size)

    def forward(self, x):
        return self.linear2(self.activation(self.linear1(x)))


class PlainBlock(nn.Module):
    def __init__(self, d, H, I, causal=True):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d)
        self.attn = PlainSelfAttention(d, H, causal)
        self.ln_2 = nn.LayerNorm(d)
        self.mlp = PlainMLP(d, I)

    def forward(self, x):
        x = x + self.attn(self.ln_1(x))
        return x + self.mlp(self.ln_2(x))


class PlainGPT2Stack(nn.Module):
    """GPT-2-shaped block stack on hidden states [B, S, d]."""

    def __init__(self, hidden_size=1024, num_heads=16, num_layers=24, intermediate_size=None, causal=True, seed=0):
        super().__init__()
        I = intermediate_size or 4 * hidden_size
        g = torch.Generator().manual_seed(seed)
        self.h = nn.ModuleList([PlainBlock(hidden_size, num_heads, I, causal) for _ in range(num_layers)])
        self.ln_f = nn.LayerNorm(hidden_size)
        with torch.no_grad():  # weights N(0, 0.02), biases 0 (flash_attention.py:534-542)
            for m in self.modules():
                if isinstance(m, nn.Linear):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.02)
                    m.bias.zero_()

    def forward(self, x):
        for blk in self.h:
            x = blk(x)
        return self.ln_f(x)


class BasicInferenceRunner:
    """warmup / run_inference of baseline/inference.py:616-713 for a `model(inputs)` module."""

    def __init__(self, model: nn.Module, device: str = "cpu"):
        self.model, self.device, self.metrics = model.to(device), device, {}

    def _forward(self, inputs: Any, **kwargs) -> Any:
        return self.model(inputs, **kwargs)

    def warmup(self, inputs: Any, iterations: int = 10) -> None:
        with torch.no_grad():
            self.model.eval()
            for _ in range(iterations):
                self._forward(inputs)

    def run_inference(self, inputs: Any, **kwargs) -> Tuple[Any, Dict[str, float]]:
        self.model.eval()
        t0 = time.perf_counter()
        with torch.no_grad():
            out = self._forward(inputs, **kwargs)
        self.metrics = {"total_time_ms": (time.perf_counter() - t0) * 1000.0}
        return out, self.metrics


def time_cpu_baseline(hidden_size=1024, num_heads=16, num_layers=1, batch=1, seq_len=4096,
                      warmup=1, iters=2, seed=0) -> Dict[str, float]:
    """tokens/s of the plain fp32 stack on the host cores through BasicInferenceRunner."""
    torch.manual_seed(seed)
    model = PlainGPT2Stack(hidden_size, num_heads, num_layers, seed=seed)
    x = torch.randn(batch, seq_len, hidden_size)
    runner = BasicInferenceRunner(model, "cpu")
    runner.warmup(x, warmup)
    ts = []
    for _ in range(iters):
        _, m = runner.run_inference(x)
        ts.append(m["total_time_ms"])
    avg_s = sum(ts) / len(ts) / 1000.0
    return {"tokens_per_s": batch * seq_len / avg_s, "avg_latency_s": avg_s,
            "threads": torch.get_num_threads(), "layers": num_layers}
