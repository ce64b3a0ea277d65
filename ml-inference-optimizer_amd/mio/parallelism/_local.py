"""Per-rank compute used by the parallel wrappers: the HIP kernels.

The distributed schedules (tensor-parallel sharding + all-reduce, ring / mesh K-V exchange + (o, lse)
carry) are backend-agnostic; what runs on each rank between two communication steps goes through the
functions below.  In the product they are the HIP kernels and nothing else.  The CPU (gloo)
schedule tests replace them with checker implementations to exercise the communication logic without
a GPU (tests/test_parallel_gloo.py)."""
from __future__ import annotations

from .. import ops


import weakref

_blocked = {}  # id(weight tensor) -> (weakref to it, key, blocked copy): the 256x256-tile GEMMs take blocked weights


def _blocked_weight(w):
    # keyed like CastCache.get_blocked: module.half() / .to(device) / `param.data = loaded` keep id() and _version
    key = (w.data_ptr(), w._version, w.dtype, w.device, tuple(w.shape))
    hit = _blocked.get(id(w))
    if hit is not None and hit[0]() is w and hit[1] == key:
        return hit[2]
    if len(_blocked) > 256:
        _blocked.clear()
    wb = ops.block_weight(w.detach())
    _blocked[id(w)] = (weakref.ref(w), key, wb)
    return wb


def linear(x, weight, bias=None, activation="none", residual=None, out=None, col_scale=None):
    """F.linear (+ activation, + residual) on the MFMA GEMM; at sizes that run the 256x256-tile kernels the weight goes in
    the blocked layout (repacked once per parameter version).  col_scale = (lo, hi, value): see ops.gemm_bias_act (only
    where k_prescale_ok() said yes)."""
    N, K = weight.shape
    M = x.numel() // K
    wb = _blocked_weight(weight) if (K % 32 == 0 and ops.blocked_weight_ok(M, N, K, activation)) else None
    return ops.gemm_bias_act(x, weight, bias, activation, residual=residual, out=out, w_blocked=wb, col_scale=col_scale)


def k_prescale_ok(B, Sq, H, D, M, N, K, carry=True, row_stride=None):
    """True iff the projection that produces K ([M, K] x [N, K]^T) can scale columns in its epilogue AND the attention launches
    (Sq query rows per launch, head dim D; carry: with the ring's (o_acc, lse) state) take pre-scaled K."""
    rs = H * D if row_stride is None else row_stride
    return ops.col_scale_ok(M, N, K) and ops.fa3_k_prescaled_ok(B, Sq, Sq, H, D, rs, rs, carry=carry)


def attention_step(q, k, v, **kw):
    """One kernel launch of tiled attention with optional (o_acc, lse) carry; see ops.fa3_fwd."""
    return ops.fa3_fwd(q, k, v, **kw)


def layernorm(x, weight, bias=None, eps=1e-5):
    """Row LayerNorm (the pre-LN block's `module(ln(x))` when a converted block is called with pre_norm=)."""
    w = weight if weight.dtype == x.dtype else weight.to(x.dtype)
    b = bias if (bias is None or bias.dtype == x.dtype) else bias.to(x.dtype)
    return ops.layernorm(x, w, b, eps)
