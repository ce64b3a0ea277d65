"""Per-rank compute used by the parallel wrappers: the HIP kernels.

The distributed schedules (tensor-parallel sharding + all-reduce, ring / mesh K-V exchange + (o, lse)
carry) are backend-agnostic; what runs on each rank between two communication steps goes through the
functions below.  In the product they are the HIP kernels and nothing else.  The CPU (gloo)
schedule tests replace them with checker implementations to exercise the communication logic without
a GPU (tests/test_parallel_gloo.py)."""
from __future__ import annotations

from .. import ops


def linear(x, weight, bias=None, activation="none", residual=None, out=None):
    return ops.gemm_bias_act(x, weight, bias, activation, residual=residual, out=out)


def attention_step(q, k, v, **kw):
    """One kernel launch of tiled attention with optional (o_acc, lse) carry; see ops.fa3_fwd."""
    return ops.fa3_fwd(q, k, v, **kw)


def layernorm(x, weight, bias=None, eps=1e-5):
    """Row LayerNorm (the pre-LN block's `module(ln(x))` when a converted block is called with pre_norm=)."""
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if (bias is None or bias.dtype == x.dtype) else bias.to(x.dtype)
    return ops.layernorm(x, w, b, eps)
