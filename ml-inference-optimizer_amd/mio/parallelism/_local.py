"""Per-rank compute used by the parallel wrappers: the HIP kernels.

The distributed schedules (tensor-parallel sharding + all-reduce, ring / mesh K-V exchange + (o, lse)
carry) are backend-agnostic; what runs on each rank between two communication steps goes through the
functions below.  In the product they are the HIP kernels and nothing else.  The CPU (gloo)
schedule tests replace them with checker implementations to exercise the communication logic without
a GPU (tests/test_parallel_gloo.py)."""
from __future__ import annotations

from .. import ops


_blocked = {}  # (data_ptr, version, shape, dtype) -> blocked copy of a weight (col_scale launches take blocked weights)


def _blocked_weight(w):
    key = (w.data_ptr(), w._version, tuple(w.shape), w.dtype)
    wb = _blocked.get(key)
    if wb is None:
        if len(_blocked) > 64:
            _blocked.clear()
        wb = _blocked[key] = ops.block_weight(w.detach())
    return wb


def linear(x, weight, bias=None, activation="none", residual=None, out=None, col_scale=None):
    """col_scale = (lo, hi, value): see ops.gemm_bias_act (only where k_prescale_ok() said yes)."""
    if col_scale is not None:
        return ops.gemm_bias_act(x, weight, bias, activation, out=out, w_blocked=_blocked_weight(weight), col_scale=col_scale)
    return ops.gemm_bias_act(x, weight, bias, activation, residual=residual, out=out)


def k_prescale_ok(B, Sq, H, D, M, N, K):
    """True iff the K projection ([M, K] x [N, K]^T) can scale its columns in its epilogue AND the ring's attention launches
    ((o_acc, lse) carry, Sq query rows per launch, head dim D) take pre-scaled K."""
    return ops.col_scale_ok(M, N, K) and ops.fa3_k_prescaled_ok(B, Sq, Sq, H, D, H * D, H * D, carry=True)


def attention_step(q, k, v, **kw):
    """One kernel launch of tiled attention with optional (o_acc, lse) carry; see ops.fa3_fwd."""
    return ops.fa3_fwd(q, k, v, **kw)


def layernorm(x, weight, bias=None, eps=1e-5):
    """Row LayerNorm (the pre-LN block's `module(ln(x))` when a converted block is called with pre_norm=)."""
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if (bias is None or bias.dtype == x.dtype) else bias.to(x.dtype)
    return ops.layernorm(x, w, b, eps)
