"""Checker implementations of mio.parallelism._local for the CPU (gloo) schedule tests.

They reproduce the CONTRACT of the two HIP entry points (ops.gemm_bias_act, ops.fa3_fwd incl. the
(o_acc, lse) carry) with the CPU oracle, so the distributed schedules can be exercised without a GPU.
Test-only: the product never imports this."""
import torch

import oracle
from oracle.mlp import gelu_tanh

_ACTS = {"none": lambda x: x, "gelu": gelu_tanh, "gelu_erf": torch.nn.functional.gelu,
         "relu": torch.relu, "silu": torch.nn.functional.silu}


def linear(x, weight, bias=None, activation="none", residual=None, out=None, col_scale=None):
    y = torch.nn.functional.linear(x.double(), weight.double(), None if bias is None else bias.double())
    y = _ACTS[activation](y)
    if col_scale is not None:  # (lo, hi, value): scaled before the one rounding to the storage dtype
        lo, hi, val = col_scale
        if lo % 128 or hi % 128:  # ops.gemm_bias_act / mio_gemm_bias_act_bw_cs refuse such a range
            raise ValueError(f"col_scale range [{lo}, {hi}) must be multiples of 128")
        y[..., lo:hi] = y[..., lo:hi] * val
    if residual is not None:
        y = y + residual.double()
    y = y.to(x.dtype)
    if out is not None:
        out.copy_(y)
        return out
    return y


def attention_step(q, k, v, *, layout="bshd", causal=False, softmax_scale=None, keep_mask=None, additive_mask=None,
                   return_lse=False, out=None, o_acc=None, lse=None, carry_in=False, write_out=True, q_offset=0,
                   k_offset=0, k_prescaled=False):
    if k_prescaled:  # k already holds K * softmax_scale * log2(e): scores = q . k~ in base 2
        import math
        softmax_scale = math.log(2.0)
    if layout == "bhsd":
        qs, ks, vs = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    else:
        qs, ks, vs = q, k, v
    o_new, lse_new = oracle.attention_with_lse(qs, ks, vs, mask=keep_mask, causal=causal, softmax_scale=softmax_scale,
                                               additive_mask=additive_mask, q_offset=q_offset, k_offset=k_offset)
    if carry_in:
        o_new, lse_new = oracle.merge_attention_states(o_acc, lse, o_new, lse_new)
    if o_acc is not None:
        o_acc.copy_(o_new.float())
        lse.copy_(lse_new.float())
    res = None
    if write_out:
        res = o_new.to(q.dtype)
        if layout == "bhsd":
            res = res.permute(0, 2, 1, 3)
        if out is not None:
            out.copy_(res)
            res = out
    if return_lse:
        return res, lse_new.float()
    return res


def layernorm(x, weight, bias=None, eps=1e-5):
    return oracle.layernorm(x, weight, bias, eps).to(x.dtype)


def k_prescale_ok(B, Sq, H, D, M, N, K, carry=True, row_stride=None):
    return D <= 64  # (the HIP rule also wants Sq > 128 and a 256-tile GEMM shape; the schedule tests run smaller)


def install(monkeypatch=None):
    from mio.parallelism import _local
    if monkeypatch is not None:
        monkeypatch.setattr(_local, "linear", linear)
        monkeypatch.setattr(_local, "attention_step", attention_step)
        monkeypatch.setattr(_local, "layernorm", layernorm)
        monkeypatch.setattr(_local, "k_prescale_ok", k_prescale_ok)
    else:
        _local.linear = linear
        _local.attention_step = attention_step
        _local.layernorm = layernorm
        _local.k_prescale_ok = k_prescale_ok
