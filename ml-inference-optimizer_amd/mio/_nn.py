"""Small host-side helpers shared by the nn.Module wrappers (precision casting, Linear via the HIP GEMM)."""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import ops

_PRECISION = {"fp16": torch.float16, "bf16": torch.bfloat16}


def compute_dtype(precision: str, x: torch.Tensor) -> torch.dtype:
    """dtype the HIP kernels run in for a module configured with `precision`.

    The reference casts to `config.precision` before its Triton launch
    (flash_attention.py:176-198, fused_mlp.py:106-123).  fp16/bf16 map directly; "fp32" means
    "keep the tensor's dtype" there, which the MFMA kernels cannot do: a 16-bit input is used as
    is, an fp32 input raises (no silent down-cast of an fp32-configured module)."""
    if precision in _PRECISION:
        return _PRECISION[precision]
    if precision == "fp32":
        if x.dtype in (torch.float16, torch.bfloat16):
            return x.dtype
        raise RuntimeError("precision='fp32' with fp32 tensors is not supported by the MFMA kernels; "
                           "configure precision='bf16' or 'fp16'")
    if precision == "fp8":
        raise RuntimeError("FP8 precision is not supported by the HIP path (the reference gates it to "
                           "Hopper, flash_attention.py:89-100)")
    raise ValueError(f"Unsupported precision mode: {precision}")


class CastCache:
    """Caches parameter copies in the compute dtype, keyed on (data_ptr, _version, dtype), so a module
    whose parameters are stored in another dtype does not re-cast them on every forward."""

    def __init__(self):
        self._c: Dict[object, Tuple[Tuple, torch.Tensor]] = {}

    def get(self, p: Optional[torch.Tensor], dtype: torch.dtype) -> Optional[torch.Tensor]:
        if p is None:
            return None
        if p.dtype == dtype and p.is_contiguous():
            return p.detach()
        key = (p.data_ptr(), p._version, dtype, p.device)
        hit = self._c.get(id(p))
        if hit is not None and hit[0] == key:
            return hit[1]
        t = p.detach().to(dtype).contiguous()
        self._c[id(p)] = (key, t)
        return t

    def get_blocked(self, p: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """The parameter in the blocked weight layout (ops.block_weight), repacked when its version changes."""
        key = (p.data_ptr(), p._version, dtype, p.device, "blocked")
        hit = self._c.get(("b", id(p)))
        if hit is not None and hit[0] == key:
            return hit[1]
        t = ops.block_weight(self.get(p, dtype))
        self._c[("b", id(p))] = (key, t)
        return t


def _get_blocked_glu(self, gate: torch.Tensor, up: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The SwiGLU gate / up parameters as ONE interleaved blocked weight (ops.block_weight_glu), repacked when either changes."""
    key = (gate.data_ptr(), gate._version, up.data_ptr(), up._version, dtype, gate.device, "blocked_glu")
    hit = self._c.get(("g", id(gate), id(up)))
    if hit is not None and hit[0] == key:
        return hit[1]
    t = ops.block_weight_glu(self.get(gate, dtype), self.get(up, dtype))
    self._c[("g", id(gate), id(up))] = (key, t)
    return t


CastCache.get_blocked_glu = _get_blocked_glu


def _get_ln_folded(self, lin: nn.Linear, ln: nn.LayerNorm, dtype: torch.dtype):
    """(blocked gamma-scaled, row-centred weight, beta-folded bias) of a projection behind LayerNorm `ln` (ops.ln_fold_weight),
    prepared once per version of the four parameters involved."""
    ps = (lin.weight, lin.bias, ln.weight, ln.bias)
    key = tuple((None if t is None else (t.data_ptr(), t._version)) for t in ps) + (dtype, lin.weight.device, "ln_fold")
    slot = ("f", id(lin.weight), id(ln.weight))
    hit = self._c.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    t = ops.ln_fold_weight(self.get(lin.weight, dtype), self.get(ln.weight, dtype), self.get(ln.bias, dtype), self.get(lin.bias, dtype))
    self._c[slot] = (key, t)
    return t


CastCache.get_ln_folded = _get_ln_folded


def _get_ln_folded_glu(self, gate: nn.Linear, up: nn.Linear, ln: nn.LayerNorm, dtype: torch.dtype):
    """The SwiGLU pair behind LayerNorm `ln`: (interleaved blocked weight of the two folded weights, up bias', gate bias')."""
    ps = (gate.weight, gate.bias, up.weight, up.bias, ln.weight, ln.bias)
    key = tuple((None if t is None else (t.data_ptr(), t._version)) for t in ps) + (dtype, up.weight.device, "ln_fold_glu")
    slot = ("fg", id(gate.weight), id(up.weight), id(ln.weight))
    hit = self._c.get(slot)
    if hit is not None and hit[0] == key:
        return hit[1]
    lw, lb = self.get(ln.weight, dtype), self.get(ln.bias, dtype)
    wg, bg = ops.ln_fold_weight(self.get(gate.weight, dtype), lw, lb, self.get(gate.bias, dtype), blocked=False)
    wu, bu = ops.ln_fold_weight(self.get(up.weight, dtype), lw, lb, self.get(up.bias, dtype), blocked=False)
    t = (ops.block_weight_glu(wg, wu), bu, bg)
    self._c[slot] = (key, t)
    return t


CastCache.get_ln_folded_glu = _get_ln_folded_glu


class ResidualStream:
    """The residual stream between two sub-layers as the folded kernels hand it on (ops.gemm_ln): `blocked` is the
    [ceil(M/256)*256, d] tensor in the blocked activation layout, `stats` the (sum, sum of squares) row statistics its
    producer wrote beside it ([d/256, ceil(M/256)*256, 2] fp32), `shape` the logical (B, S, d).  A sub-layer that takes a
    ResidualStream normalises inside its first GEMM's read-out and reads the residual from `blocked`; no LayerNorm launch."""
    __slots__ = ("blocked", "stats", "shape")

    def __init__(self, blocked: torch.Tensor, stats: torch.Tensor, shape: Tuple[int, int, int]):
        self.blocked, self.stats, self.shape = blocked, stats, tuple(shape)

    @property
    def dtype(self):
        return self.blocked.dtype

    @property
    def device(self):
        return self.blocked.device

    def dense(self) -> torch.Tensor:
        """Row-major [B, S, d] copy (tests / debugging: a strided view copied by torch, not a kernel of this package)."""
        B, S, d = self.shape
        mp = self.blocked.shape[0]
        return self.blocked.view(mp // 256, d // 32, 256, 32).permute(0, 2, 1, 3).reshape(mp, d)[:B * S].reshape(B, S, d)


def linear(x: torch.Tensor, lin: nn.Linear, cache: CastCache, dtype: torch.dtype, activation: str = "none",
           residual: Optional[torch.Tensor] = None, col_scale=None, x_blocked_shape=None) -> torch.Tensor:
    """F.linear(x, W, b) (+ activation, + residual) on the MFMA GEMM.  At sizes that run the 256x256-tile kernels the
    weight is handed over in the blocked layout (repacked once per parameter version, cached next to the cast copy)."""
    w = cache.get(lin.weight, dtype)
    N, K = w.shape
    M = x.numel() // K if x_blocked_shape is None else int(math.prod(x_blocked_shape[:-1]))
    wb = cache.get_blocked(lin.weight, dtype) if (K % 32 == 0 and ops.blocked_weight_ok(M, N, K, activation)) else None
    return ops.gemm_bias_act(x, w, cache.get(lin.bias, dtype), activation, residual=residual, w_blocked=wb,
                             col_scale=col_scale, x_blocked_shape=x_blocked_shape)


def prenorm_linear(x: torch.Tensor, ln: nn.LayerNorm, lin: nn.Linear, cache: CastCache, dtype: torch.dtype,
                   activation: str = "none", residual: Optional[torch.Tensor] = None, col_scale=None) -> torch.Tensor:
    """lin(ln(x)) (+ activation, + residual).  At sizes that run the 256x256-tile kernels LayerNorm writes its output in
    the blocked activation layout, so the GEMM's K-tile fetches are contiguous on both operands."""
    w = cache.get(lin.weight, dtype)
    N, K = w.shape
    M = x.numel() // K
    lw, lb = cache.get(ln.weight, dtype), cache.get(ln.bias, dtype)
    if K % 32 == 0 and not ops.NO_BLOCKED_X and ops.blocked_weight_ok(M, N, K, activation):
        xb = ops.layernorm(x, lw, lb, ln.eps, out_blocked=True)
        return ops.gemm_bias_act(xb, w, cache.get(lin.bias, dtype), activation, residual=residual,
                                 w_blocked=cache.get_blocked(lin.weight, dtype), x_blocked_shape=tuple(x.shape),
                                 col_scale=col_scale)
    return linear(ops.layernorm(x, lw, lb, ln.eps), lin, cache, dtype, activation, residual, col_scale=col_scale)
