"""Functional entry points of the hot path: thin argument checking + one C-ABI call each.

These mirror the reference's functional launch wrappers (same positional/keyword signatures,
same exception types for the same conditions):
  flash_attention           <- triton_flash_attention        (kernels/triton/flash_attention_kernels.py:1150-1358)
  ring_attention_forward    <- triton_ring_attention_forward (kernels/triton/attention_kernels.py:909-1005)
  fused_mlp                 <- triton_fused_mlp              (kernels/triton/mlp_kernels.py:648-756)
  layernorm                 <- triton_layernorm              (kernels/triton/layernorm_kernels.py:191-276)
  paged_attention_forward   <- triton_paged_attention_forward(kernels/triton/attention_kernels.py:1206-1311)
  reshape_and_cache         <- triton_reshape_and_cache      (kernels/triton/attention_kernels.py:1314-1407)
There is no fallback path: a non-zero return from the library raises RuntimeError.
"""
from __future__ import annotations

import os

import ctypes as C
import math
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import lib, check

_ACT = {
    "none": _lib.ACT_NONE,
    "gelu": _lib.ACT_GELU_TANH,       # the Triton kernel's GELU is the tanh form (mlp_kernels.py:144-161)
    "gelu_tanh": _lib.ACT_GELU_TANH,
    "gelu_new": _lib.ACT_GELU_TANH,
    "gelu_erf": _lib.ACT_GELU_ERF,
    "relu": _lib.ACT_RELU,
    "silu": _lib.ACT_SILU,
    "swish": _lib.ACT_SILU,
    "swiglu": _lib.ACT_SWIGLU,
}


def _dtype_id(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.MIO_BF16
    if t.dtype == torch.float16:
        return _lib.MIO_FP16
    raise ValueError(f"HIP kernels compute in bf16 or fp16, got {t.dtype}")


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _need_cuda(*ts: Optional[torch.Tensor]) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise ValueError("HIP kernels require input tensors to be on a CUDA (ROCm) device.")


def _vec_ok(t: Optional[torch.Tensor], n: int, dtype: torch.dtype, what: str) -> None:
    """A per-column operand (bias, LayerNorm weight): 1-D, contiguous, n elements, the activations' dtype and device.
    The kernels read n elements of the activation dtype from the raw pointer, so anything else would be misread."""
    if t is None:
        return
    if not t.is_cuda:
        raise ValueError("HIP kernels require input tensors to be on a CUDA (ROCm) device.")
    if t.dim() != 1 or t.numel() != n or not t.is_contiguous():
        raise ValueError(f"{what} must be a contiguous 1-D tensor of {n} elements, got shape {tuple(t.shape)}")
    if t.dtype != dtype:
        raise ValueError(f"{what} must have the activation dtype {dtype}, got {t.dtype}")


def _res_ok(r: Optional[torch.Tensor], numel: int, dtype: torch.dtype) -> None:
    if r is None:
        return
    if not r.is_cuda:
        raise ValueError("HIP kernels require input tensors to be on a CUDA (ROCm) device.")
    if r.dtype != dtype:
        raise ValueError(f"residual must have the activation dtype {dtype}, got {r.dtype}")
    if r.numel() != numel:
        raise ValueError(f"residual has {r.numel()} elements, the output has {numel}")


def _rows16(t: torch.Tensor) -> torch.Tensor:
    """Make the last dim contiguous and every other stride a multiple of 8 elements."""
    if t.stride(-1) != 1 or any(s % 8 for s in t.stride()[:-1]) or t.data_ptr() % 16:
        return t.contiguous()
    return t


# ------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------
def fa3_fwd(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    *,
    layout: str = "bshd",
    causal: bool = False,
    softmax_scale: Optional[float] = None,
    keep_mask: Optional[torch.Tensor] = None,
    additive_mask: Optional[torch.Tensor] = None,
    return_lse: bool = False,
    out: Optional[torch.Tensor] = None,
    o_acc: Optional[torch.Tensor] = None,
    lse: Optional[torch.Tensor] = None,
    carry_in: bool = False,
    write_out: bool = True,
    q_offset: int = 0,
    k_offset: int = 0,
    k_prescaled: bool = False,
    out_blocked: bool = False,
):
    """One launch of the tiled attention kernel.

    layout "bshd": q [B,Sq,H,D], k/v [B,Sk,Hkv,D] (flash); "bhsd": head-major (ring, a9).
    keep_mask / additive_mask: 4-D, broadcastable to [B,H,Sq,Sk] (size-1 dims broadcast).
    Ring carry: o_acc fp32 [B,Sq,H,D] + lse fp32 [B,H,Sq]; carry_in continues from that state.
    k_prescaled: k already holds K * softmax_scale * log2(e), scaled in fp32 before its rounding to 16 bits
    (gemm_bias_act(col_scale=...)); only where fa3_k_prescaled_ok() says so -- ValueError otherwise.
    out_blocked (k_prescaled launches, layout "bshd", where fa3_o_blocked_ok() says so): the output is returned as a
    [ceil(B*Sq/256)*256, H*D] tensor in the blocked activation layout (include/mio_hip.h) for a following
    gemm_bias_act(..., x_blocked_shape=(B, Sq, H*D)) -- the output projection then fetches contiguous K-tiles.
    Returns out (same layout as q) or (out, lse) if return_lse.
    """
    _need_cuda(q, k, v)
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise ValueError(f"Expected 4D tensors for q, k, v but got shapes: q={q.shape}, k={k.shape}, v={v.shape}")
    if layout not in ("bshd", "bhsd"):
        raise ValueError(f"unknown layout {layout}")
    if k.dtype != q.dtype or v.dtype != q.dtype:
        raise ValueError("q, k, v must have the same dtype")
    dt = _dtype_id(q)
    q, k, v = _rows16(q), _rows16(k), _rows16(v)
    si, hi = (1, 2) if layout == "bshd" else (2, 1)
    B, Sq, H, D = q.shape[0], q.shape[si], q.shape[hi], q.shape[3]
    Sk, Hkv = k.shape[si], k.shape[hi]
    if k.shape[0] != B or v.shape != k.shape or k.shape[3] != D:
        raise ValueError(f"incompatible q/k/v shapes: q={q.shape}, k={k.shape}, v={v.shape}")
    if H % Hkv != 0:
        raise ValueError(f"num_heads {H} must be a multiple of num_kv_heads {Hkv}")
    if D % 8 != 0 or D > 128:
        raise ValueError(f"head_dim must be a multiple of 8 and <= 128, got {D}")
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else float(softmax_scale)
    if not (scale > 0.0):
        raise ValueError("softmax_scale must be positive")
    if keep_mask is not None and additive_mask is not None:
        raise ValueError("give either keep_mask or additive_mask, not both")

    p = _lib.FaParams()
    if out_blocked:
        if layout != "bshd" or not write_out or out is not None or not k_prescaled:
            raise ValueError("out_blocked needs layout 'bshd', k_prescaled, write_out and no out= tensor")
        out = torch.empty((B * Sq + 255) // 256 * 256, H * D, dtype=q.dtype, device=q.device)
    elif write_out:
        if out is None:
            out = torch.empty_like(q, memory_format=torch.contiguous_format)
        elif out.shape != q.shape or out.dtype != q.dtype or out.stride(-1) != 1:
            raise ValueError("out must match q in shape/dtype with a contiguous last dim")
    else:
        out = None
        if o_acc is None:
            raise ValueError("write_out=False needs o_acc")
    if o_acc is not None:
        if o_acc.dtype != torch.float32 or tuple(o_acc.shape) != (B, Sq, H, D) or not o_acc.is_contiguous():
            raise ValueError("o_acc must be contiguous fp32 [B,Sq,H,D]")
        if lse is None:
            raise ValueError("o_acc needs an lse buffer")
    if (return_lse or o_acc is not None) and lse is None:
        lse = torch.empty(B, H, Sq, dtype=torch.float32, device=q.device)
    if lse is not None and (lse.dtype != torch.float32 or tuple(lse.shape) != (B, H, Sq) or not lse.is_contiguous()):
        raise ValueError("lse must be contiguous fp32 [B,H,Sq]")
    if carry_in and o_acc is None:
        raise ValueError("carry_in needs o_acc and lse")

    mask, kind = None, _lib.MASK_NONE
    if keep_mask is not None:
        mask, kind = keep_mask, _lib.MASK_KEEP_U8
        if mask.dtype != torch.uint8:
            mask = (mask != 0).to(torch.uint8)
    elif additive_mask is not None:
        mask, kind = additive_mask.to(torch.float32), _lib.MASK_ADD_F32
    if mask is not None:
        _need_cuda(mask)
        if mask.dim() != 4:
            raise ValueError(f"Unsupported mask shape: {tuple(mask.shape)}")
        for dim, full in zip(mask.shape, (B, H, Sq, Sk)):
            if dim not in (1, full):
                raise ValueError(f"mask shape {tuple(mask.shape)} does not broadcast to {(B, H, Sq, Sk)}")
        for i in range(4):
            p.mask_stride[i] = 0 if mask.shape[i] == 1 else mask.stride(i)

    def _st(dst, t):
        dst[0], dst[1], dst[2] = t.stride(0), t.stride(si), t.stride(hi)

    _st(p.q_stride, q)
    _st(p.k_stride, k)
    _st(p.v_stride, v)
    if out is not None and not out_blocked:
        _st(p.o_stride, out)
    p.q, p.k, p.v = q.data_ptr(), k.data_ptr(), v.data_ptr()
    p.o, p.lse, p.o_acc, p.mask = _ptr(out), _ptr(lse), _ptr(o_acc), _ptr(mask)
    p.B, p.Sq, p.Sk, p.H, p.Hkv, p.D = B, Sq, Sk, H, Hkv, D
    p.dtype, p.causal, p.mask_kind, p.carry_in = dt, int(bool(causal)), kind, int(bool(carry_in))
    p.q_offset, p.k_offset, p.softmax_scale = int(q_offset), int(k_offset), scale
    if k_prescaled:
        if not lib.mio_fa3_k_prescaled_ok(C.byref(p)):
            raise ValueError("k_prescaled is only supported for head_dim <= 96 (<= 64 with the (o_acc, lse) carry), no mask, "
                             "Sq > 128")
        p.k_prescaled = 1
    if out_blocked:
        if not lib.mio_fa3_o_blocked_ok(C.byref(p)):
            raise ValueError("out_blocked is only supported for k_prescaled launches without carry, head_dim <= 64, "
                             "(H * D) % 32 == 0")
        p.o_blocked = 1
    check(lib.mio_fa3_fwd(C.byref(p), _stream()))
    if return_lse:
        return out, lse
    return out


def fa3_k_prescaled_ok(B: int, Sq: int, Sk: int, H: int, D: int, k_row_stride: int, v_row_stride: int,
                       carry: bool = False) -> bool:
    """True iff fa3_fwd(..., k_prescaled=True) is available for a launch of this geometry without a mask (carry: with the
    (o_acc, lse) ring carry)."""
    return (D <= (64 if carry else 96) and Sq > 128 and Sk * k_row_stride * 2 < (1 << 32)
            and Sk * v_row_stride * 2 < (1 << 32))


def fa3_o_blocked_ok(B: int, Sq: int, Sk: int, H: int, D: int, k_row_stride: int, v_row_stride: int) -> bool:
    """True iff fa3_fwd(..., k_prescaled=True, out_blocked=True) is available for this geometry."""
    return (not NO_BLOCKED_X and fa3_k_prescaled_ok(B, Sq, Sk, H, D, k_row_stride, v_row_stride) and D <= 64
            and (H * D) % 32 == 0)


def _canon_mask4(mask: torch.Tensor) -> torch.Tensor:
    """[B,S] / [B,1,S] / [B,S,S] / 4-D -> 4-D, as flash_attention_kernels.py:1232-1250."""
    if mask.dim() == 2:
        return mask[:, None, None, :]
    if mask.dim() == 3:
        return mask[:, :, None, :] if mask.shape[1] == 1 else mask[:, None, :, :]
    if mask.dim() == 4:
        return mask
    raise ValueError(f"Unsupported mask shape: {mask.shape}")


def flash_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
    causal: bool = False,
    softmax_scale: Optional[float] = None,
    dropout_p: float = 0.0,
    return_softmax: bool = False,
    block_size: int = 128,
):
    """Drop-in for triton_flash_attention (flash_attention_kernels.py:1150-1358), q/k/v [B,S,H,D].

    mask: keep-mask (nonzero = attend; masked scores := -1e9, :257-273) of shape [B,S], [B,1,S],
    [B,S,S] or 4-D.  block_size is accepted for signature compatibility; the HIP kernel's tile
    (128 queries x 64 keys) is fixed.  return_softmax / dropout_p > 0 are not computed by the
    fused kernel: NotImplementedError (the reference's autograd path punts the same way, :1044-1046).
    """
    if q.dim() != 4 or k.dim() != 4 or v.dim() != 4:
        raise ValueError(f"Expected 4D tensors for q, k, v but got shapes: q={q.shape}, k={k.shape}, v={v.shape}")
    if return_softmax:
        raise NotImplementedError("return_softmax=True is not supported by the fused HIP kernel")
    if dropout_p and dropout_p > 0.0:
        raise NotImplementedError("attention dropout (training) is not supported by the inference kernel")
    keep = None
    if mask is not None:
        keep = _canon_mask4(mask.to(q.device))
    return fa3_fwd(q, k, v, layout="bshd", causal=causal, softmax_scale=softmax_scale, keep_mask=keep)


def ring_attention_forward(
    query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
    k_prescaled: bool = False,
) -> torch.Tensor:
    """Drop-in for triton_ring_attention_forward (attention_kernels.py:909-1005; fallback :1520-1591):
    q/k/v [B,H,S,D] head-major, additive mask [B,1|H,Sq,Sk]; returns [B,Sq,H*D].
    k_prescaled (not in the reference): see fa3_fwd."""
    if query.dim() != 4:
        raise ValueError(f"Expected 4D tensors, got {query.shape}")
    B, H, Sq, D = query.shape
    out = torch.empty(B, Sq, H, D, dtype=query.dtype, device=query.device)
    # write straight into the [B,Sq,H*D] result: give the kernel a head-major VIEW of it
    fa3_fwd(query, key, value, layout="bhsd", additive_mask=attention_mask, out=out.permute(0, 2, 1, 3),
            k_prescaled=k_prescaled)
    return out.view(B, Sq, H * D)


def attn_merge(o_a, lse_a, o_b, lse_b, out: Optional[torch.Tensor] = None):
    """(o_a, lse_a) <- merge((o_a, lse_a), (o_b, lse_b)); fp32 states [B,Sq,H,D] / [B,H,Sq]."""
    _need_cuda(o_a, lse_a, o_b, lse_b)
    B, Sq, H, D = o_a.shape
    for t in (o_a, o_b, lse_a, lse_b):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("merge states must be contiguous fp32")
    dt = _lib.MIO_BF16 if out is None else _dtype_id(out)
    if out is not None and (tuple(out.shape) != (B, Sq, H, D) or not out.is_contiguous()):
        raise ValueError("out must be contiguous [B,Sq,H,D]")
    check(lib.mio_attn_merge(o_a.data_ptr(), lse_a.data_ptr(), o_b.data_ptr(), lse_b.data_ptr(), _ptr(out),
                             B, Sq, H, D, dt, _stream()))
    return o_a, lse_a


# ------------------------------------------------------------------------------------------------
# GEMM / FusedMLP / LayerNorm
# ------------------------------------------------------------------------------------------------
_NO_BLOCKED_W = os.environ.get("MIO_NO_BLOCKED_W", "0") == "1"
NO_BLOCKED_X = os.environ.get("MIO_NO_BLOCKED_X", "0") == "1"  # A/B runs: LayerNorm keeps writing the plain layout


def block_weight(w: torch.Tensor) -> torch.Tensor:
    """One-time repack of a Linear weight [N, K] (K % 32 == 0) into the blocked layout of include/mio_hip.h
    (contiguous 16 KiB K-tiles): returns a [ceil(N/256)*256, K] tensor holding the blocked bytes."""
    _need_cuda(w)
    if w.dim() != 2 or w.shape[1] % 32 != 0:
        raise ValueError(f"block_weight needs a 2-D weight with K % 32 == 0, got {tuple(w.shape)}")
    w = _rows16(w)
    N, K = w.shape
    wb = torch.empty((N + 255) // 256 * 256, K, dtype=w.dtype, device=w.device)
    assert wb.numel() * wb.element_size() == lib.mio_weight_blocked_bytes(N, K)
    check(lib.mio_weight_block(w.data_ptr(), w.stride(0), wb.data_ptr(), N, K, _dtype_id(w), _stream()))
    return wb


def block_weight_glu(w_gate: torch.Tensor, w_up: torch.Tensor) -> torch.Tensor:
    """One-time repack of the SwiGLU gate / up weights [I, K] (K % 32 == 0) into ONE blocked weight whose 256-row tiles
    interleave 32 gate rows and the 32 up rows of the same output columns per wave slice (include/mio_hip.h):
    returns a [ceil(I/128)*256, K] tensor.  fused_mlp(..., activation="swiglu", fc1_blocked=<this>)."""
    _need_cuda(w_gate, w_up)
    if w_gate.dim() != 2 or w_gate.shape != w_up.shape or w_gate.shape[1] % 32 != 0 or w_gate.dtype != w_up.dtype:
        raise ValueError(f"block_weight_glu needs two 2-D weights of one shape and dtype with K % 32 == 0, got "
                         f"{tuple(w_gate.shape)} / {tuple(w_up.shape)}")
    w_gate, w_up = _rows16(w_gate), _rows16(w_up)
    if w_gate.stride(0) != w_up.stride(0):
        w_gate, w_up = w_gate.contiguous(), w_up.contiguous()
    I, K = w_gate.shape
    wb = torch.empty((I + 127) // 128 * 256, K, dtype=w_gate.dtype, device=w_gate.device)
    assert wb.numel() * wb.element_size() == lib.mio_weight_blocked_glu_bytes(I, K)
    check(lib.mio_weight_block_glu(w_gate.data_ptr(), w_up.data_ptr(), w_gate.stride(0), wb.data_ptr(), I, K,
                                   _dtype_id(w_gate), _stream()))
    return wb


def blocked_weight_ok(M: int, N: int, K: int, activation: str = "none") -> bool:
    """True iff a GEMM of this shape runs a kernel that takes blocked weights (w_blocked= below).
    MIO_NO_BLOCKED_W=1 (A/B runs) keeps the modules on the plain weights."""
    return not _NO_BLOCKED_W and bool(lib.mio_gemm_blocked_weight_ok(M, N, K, _ACT.get(activation, _lib.ACT_NONE)))


def fused_mlp_blocked_weight_ok(M: int, d: int, I: int, activation: str) -> bool:
    """True iff ops.fused_mlp at this shape takes blocked fc1 / fc2 weights (fc1_blocked= / fc2_blocked=)."""
    return not _NO_BLOCKED_W and bool(lib.mio_fused_mlp_blocked_weight_ok(M, d, I, _ACT.get(activation, _lib.ACT_NONE)))


def col_scale_ok(M: int, N: int, K: int, activation: str = "none") -> bool:
    """True iff gemm_bias_act(..., col_scale=) is available for this shape (the persistent 256x256-tile kernel runs it)."""
    return not _NO_BLOCKED_W and bool(lib.mio_gemm_col_scale_ok(M, N, K, _ACT.get(activation, _lib.ACT_NONE)))


def gemm_bias_act(x, w, bias=None, activation: str = "none", w_gate=None, bias_gate=None, residual=None, out=None,
                  w_blocked=None, x_blocked_shape=None, col_scale=None):
    """y = act(x @ w^T + bias) (+ residual); x [..., K], w [N, K].  F.linear with a fused epilogue.
    col_scale = (lo, hi, value): output columns [lo, hi) are multiplied by value in fp32 before the rounding to the
    storage dtype (lo, hi multiples of 128; needs w_blocked, no residual, and col_scale_ok())."""
    _need_cuda(x, w)
    if activation not in _ACT:
        raise ValueError(f"Unsupported activation function: {activation}")
    act = _ACT[activation]
    dt = _dtype_id(x)
    if w.dtype != x.dtype:
        raise ValueError("x and w must have the same dtype")
    K = x.shape[-1]
    N = w.shape[0]
    if w.shape[1] != K:
        raise ValueError(f"weight shape {tuple(w.shape)} does not match input features {K}")
    _vec_ok(bias, N, x.dtype, "bias")
    if act == _lib.ACT_SWIGLU:
        _vec_ok(bias_gate, N, x.dtype, "bias_gate")
    if w_blocked is not None:
        # the blocked copy travels as a raw pointer: a stale repack (other dtype / device / shape) must not get that far
        if w_blocked.dtype != x.dtype or w_blocked.device != x.device:
            raise ValueError("w_blocked must have x's dtype and device (repack after converting the module)")
        if w_blocked.numel() != (N + 255) // 256 * 256 * K or not w_blocked.is_contiguous():
            raise ValueError(f"w_blocked has {w_blocked.numel()} elements, expected ceil(N/256)*256*K = {(N + 255) // 256 * 256 * K}")
    if col_scale is not None:
        lo, hi, val = int(col_scale[0]), int(col_scale[1]), float(col_scale[2])
        Mcs = int(math.prod(x_blocked_shape[:-1])) if x_blocked_shape is not None else x.numel() // K
        if w_blocked is None or residual is not None or not lib.mio_gemm_col_scale_ok(Mcs, N, K, act):
            raise ValueError("col_scale needs a blocked weight, no residual and a shape with col_scale_ok()")
        if lo % 128 or hi % 128 or not (0 <= lo <= hi <= N):
            raise ValueError(f"col_scale range [{lo}, {hi}) must be multiples of 128 inside [0, {N}]")
        if x_blocked_shape is not None:
            xs, ldx, xb = x, K, 1
            lead = tuple(x_blocked_shape[:-1])
        else:
            xs = _rows16(x.reshape(-1, K))
            ldx, xb = xs.stride(0), 0
            lead = tuple(x.shape[:-1])
        if out is None:
            out = torch.empty(*lead, N, dtype=x.dtype, device=x.device)
        y2 = out.view(-1, N)
        check(lib.mio_gemm_bias_act_bw_cs(xs.data_ptr(), w_blocked.data_ptr(), _ptr(bias), y2.data_ptr(), Mcs, N, K, ldx,
                                          y2.stride(0), act, dt, xb, lo, hi, val, _stream()))
        return out
    if x_blocked_shape is not None:
        # x is layernorm(..., out_blocked=True) of a tensor of shape x_blocked_shape: blocked activation layout
        M = int(math.prod(x_blocked_shape[:-1]))
        if w_blocked is None or act == _lib.ACT_SWIGLU or not lib.mio_gemm_blocked_weight_ok(M, N, K, act):
            raise ValueError("a blocked activation operand needs a blocked weight and a shape with blocked_weight_ok()")
        if out is None:
            out = torch.empty(*x_blocked_shape[:-1], N, dtype=x.dtype, device=x.device)
        y2 = out.view(-1, N)
        _res_ok(residual, M * N, x.dtype)
        r2 = None if residual is None else _rows16(residual.reshape(-1, N))
        check(lib.mio_gemm_bias_act_bw(x.data_ptr(), w_blocked.data_ptr(), _ptr(bias), _ptr(r2), y2.data_ptr(), M, N, K,
                                       K, y2.stride(0), 0 if r2 is None else r2.stride(0), act, dt, 1, _stream()))
        return out
    x2 = x.reshape(-1, K)
    x2, w = _rows16(x2), _rows16(w)
    M = x2.shape[0]
    if act == _lib.ACT_SWIGLU:
        if w_gate is None:
            raise ValueError("SwiGLU activation requires gate weights")
        w_gate = _rows16(w_gate)
    else:
        w_gate, bias_gate = None, None
    if out is None:
        out = torch.empty(*x.shape[:-1], N, dtype=x.dtype, device=x.device)
    y2 = out.view(-1, N)
    r2 = None
    if residual is not None:
        _res_ok(residual, M * N, x.dtype)
        r2 = _rows16(residual.reshape(-1, N))
    if w_blocked is not None and act != _lib.ACT_SWIGLU and lib.mio_gemm_blocked_weight_ok(M, N, K, act) and \
            x2.stride(0) * 512 < 0x7fffffff:
        check(lib.mio_gemm_bias_act_bw(x2.data_ptr(), w_blocked.data_ptr(), _ptr(bias), _ptr(r2), y2.data_ptr(), M, N, K,
                                       x2.stride(0), y2.stride(0), 0 if r2 is None else r2.stride(0), act, dt, 0, _stream()))
        return out
    check(lib.mio_gemm_bias_act(x2.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(w_gate), _ptr(bias_gate), _ptr(r2),
                                y2.data_ptr(), M, N, K, x2.stride(0), w.stride(0), y2.stride(0),
                                0 if r2 is None else r2.stride(0), act, dt, _stream()))
    return out


# ---- LayerNorm folded into the GEMMs on either side of it (mio_gemm_ln_bw; reference fused_layernorm_qkv.py:37-420) ------------
def gemm_ln_ok(M: int, N: int, K: int, activation: str = "none", fold_in: bool = False, stats_out: bool = False) -> bool:
    """True iff gemm_ln(...) runs this shape: fold_in = the projection behind a LayerNorm (ln_stats given),
    stats_out = the residual GEMM that also writes the row statistics of its output."""
    return not _NO_BLOCKED_W and bool(lib.mio_gemm_ln_ok(M, N, K, _ACT.get(activation, _lib.ACT_NONE), int(fold_in), int(stats_out)))


def ln_fold_weight(w: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], bias: Optional[torch.Tensor],
                   blocked: bool = True):
    """One-time preparation of a projection that sits behind a LayerNorm(gamma, beta): returns
    (blocked weight = gamma-scaled rows with their mean over k subtracted, bias' [N] = bias + w @ beta).  With the rows centred
    x @ w'^T equals (x - mean(x)) @ (gamma * w)^T, so gemm_ln's read-out only multiplies by rstd and adds bias'.
    blocked=False returns the row-major [N, K] weight instead (the SwiGLU pair goes through block_weight_glu afterwards)."""
    _need_cuda(w, gamma)
    dt = _dtype_id(w)
    N, K = w.shape
    _vec_ok(gamma, K, w.dtype, "gamma")
    _vec_ok(beta, K, w.dtype, "beta")
    _vec_ok(bias, N, w.dtype, "bias")
    w = _rows16(w)
    ws = torch.empty(N, K, dtype=w.dtype, device=w.device)
    bout = torch.empty(N, dtype=w.dtype, device=w.device)
    check(lib.mio_ln_fold_weight(w.data_ptr(), w.stride(0), gamma.data_ptr(), _ptr(beta), _ptr(bias), ws.data_ptr(),
                                 bout.data_ptr(), N, K, dt, _stream()))
    return (block_weight(ws) if blocked else ws), bout


def ln_stats_shape(M: int, width: int):
    return ((width + 255) // 256, (M + 255) // 256 * 256, 2)


def gemm_ln(x: torch.Tensor, w_blocked: torch.Tensor, bias: Optional[torch.Tensor], *, M: int, N: int, K: int,
            activation: str = "none", x_blocked: bool = False, residual: Optional[torch.Tensor] = None,
            res_blocked: bool = False, out_blocked: bool = False, ln_stats: Optional[torch.Tensor] = None,
            eps: float = 1e-5, stats_out: bool = False, col_scale=None, bias_gate: Optional[torch.Tensor] = None):
    """y = act(LN?(x) @ w^T + bias) (+ residual) on the 256-tile kernels with the LayerNorm folded in (module docstring of
    include/mio_hip.h, "LayerNorm folded into the GEMMs on either side of it").  Operands are [M, *] row-major 2-D tensors or,
    where the *_blocked flag says so, [ceil(M/256)*256, *] tensors in the blocked activation layout.
      ln_stats:           x is the raw residual stream, w_blocked / bias come from ln_fold_weight(...), ln_stats from the GEMM
                          that wrote x (stats_out=True);
      stats_out=True:     also returns the (sum, sum of squares) statistics of the rounded output rows.
    Returns (y, stats) -- stats is None unless stats_out."""
    _need_cuda(x, w_blocked)
    if activation not in _ACT:
        raise ValueError(f"Unsupported activation function: {activation}")
    act, dt = _ACT[activation], _dtype_id(x)
    glu = act == _lib.ACT_SWIGLU  # w_blocked = block_weight_glu(gate', up'), bias = up bias, bias_gate = gate bias, N = I
    if bias_gate is not None and not glu:
        raise ValueError("bias_gate belongs to activation='swiglu'")
    fold = ln_stats is not None
    if stats_out and residual is None:
        raise ValueError("gemm_ln: stats_out is the residual epilogue's form (give the residual)")
    if fold and residual is not None:
        raise ValueError("gemm_ln: the consumer form (ln_stats) takes no residual")
    if glu and residual is not None:
        raise ValueError("gemm_ln: the gated stage takes no residual")
    if not lib.mio_gemm_ln_ok(M, N, K, act, int(fold), int(stats_out)):
        raise ValueError("gemm_ln: this shape / activation does not take the folded kernels (gemm_ln_ok)")
    mp = (M + 255) // 256 * 256

    def _operand(t, cols, blocked, what):
        if t.dtype != x.dtype or t.device != x.device:
            raise ValueError(f"{what} must have x's dtype and device")
        if blocked:
            if t.numel() != mp * cols or not t.is_contiguous():
                raise ValueError(f"{what}: a blocked operand has ceil(M/256)*256 x {cols} contiguous elements")
            return t, cols
        t2 = _rows16(t.reshape(-1, cols))
        if t2.shape[0] != M:
            raise ValueError(f"{what}: expected [{M}, {cols}]")
        return t2, t2.stride(0)

    x2, ldx = _operand(x, K, x_blocked, "x")
    wn = (N + 127) // 128 * 256 * K if glu else (N + 255) // 256 * 256 * K
    if w_blocked.dtype != x.dtype or w_blocked.device != x.device or not w_blocked.is_contiguous() or w_blocked.numel() != wn:
        raise ValueError(f"w_blocked: expected {wn} contiguous elements of x's dtype on its device (swiglu: block_weight_glu)")
    _vec_ok(bias, N, x.dtype, "bias")
    _vec_ok(bias_gate, N, x.dtype, "bias_gate")
    r2, ldr = (None, 0) if residual is None else _operand(residual, N, res_blocked, "residual")
    slots = 0
    if fold:
        want = ln_stats_shape(M, K)
        if ln_stats.dtype != torch.float32 or tuple(ln_stats.shape) != want or not ln_stats.is_contiguous() or ln_stats.device != x.device:
            raise ValueError(f"ln_stats: expected a contiguous fp32 tensor of shape {want}")
        slots = want[0]
        if slots > 8:  # a stream wider than 2048 columns: sum the slots in groups down to what the kernel's LDS region holds
            out_slots = max(s_ for s_ in range(1, 9) if slots % s_ == 0)
            red = torch.empty((out_slots,) + want[1:], dtype=torch.float32, device=x.device)
            check(lib.mio_ln_stats_reduce(ln_stats.data_ptr(), slots, red.data_ptr(), out_slots, M, _stream()))
            ln_stats, slots = red, out_slots
    lo = hi = 0
    val = 1.0
    if col_scale is not None:
        lo, hi, val = int(col_scale[0]), int(col_scale[1]), float(col_scale[2])
        if residual is not None or lo % 128 or hi % 128 or not (0 <= lo <= hi <= N):
            raise ValueError(f"col_scale range [{lo}, {hi}) must be multiples of 128 inside [0, {N}], without a residual")
    y = torch.empty(mp if out_blocked else M, N, dtype=x.dtype, device=x.device)
    st = torch.empty(ln_stats_shape(M, N), dtype=torch.float32, device=x.device) if stats_out else None
    flags = (1 if x_blocked else 0) | (2 if out_blocked else 0) | (4 if (res_blocked and residual is not None) else 0)
    check(lib.mio_gemm_ln_bw(x2.data_ptr(), w_blocked.data_ptr(), _ptr(bias), _ptr(bias_gate), _ptr(r2), y.data_ptr(), M, N, K, ldx, N, ldr,
                             act, dt, flags, _ptr(ln_stats), slots, float(eps), _ptr(st), lo, hi, val, _stream()))
    return y, st


def _blocked_sizes_ok(fc1_blocked, fc2_blocked, x, d, I, act):
    """The blocked copies travel as raw pointers: refuse a stale repack (other dtype / device / shape)."""
    n1 = (I + 127) // 128 * 256 * d if act == _lib.ACT_SWIGLU else (I + 255) // 256 * 256 * d
    n2 = (d + 255) // 256 * 256 * I
    for t, n, what in ((fc1_blocked, n1, "fc1_blocked"), (fc2_blocked, n2, "fc2_blocked")):
        if t.dtype != x.dtype or t.device != x.device or t.numel() != n or not t.is_contiguous():
            raise ValueError(f"{what}: expected {n} contiguous elements of the input's dtype on its device "
                             f"(swiglu: block_weight_glu(gate, up)); repack after converting the module")


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
    fc1_blocked: Optional[torch.Tensor] = None,
    fc2_blocked: Optional[torch.Tensor] = None,
    x_blocked_shape=None,
) -> torch.Tensor:
    """Drop-in for triton_fused_mlp (mlp_kernels.py:648-756): fc2(act(fc1(x))), hidden [B,S,d].
    "gelu" is the tanh form like the Triton kernel (:144-161); "gelu_erf" is pytorch_fused_mlp's (:782-783)."""
    if x_blocked_shape is None and hidden_states.dim() != 3:
        raise ValueError(f"Expected 3D input tensor, got shape: {hidden_states.shape}")
    _need_cuda(hidden_states)
    if activation not in _ACT or _ACT[activation] == _lib.ACT_NONE:
        raise ValueError(f"Unsupported activation function: {activation}")
    act = _ACT[activation]
    if x_blocked_shape is not None:
        # hidden_states is layernorm(..., out_blocked=True) of a [B,S,d] tensor: blocked activation layout
        d, I = x_blocked_shape[-1], fc1_weight.shape[0]
        M = int(math.prod(x_blocked_shape[:-1]))
        _vec_ok(fc1_bias, I, hidden_states.dtype, "fc1_bias")
        _vec_ok(fc2_bias, d, hidden_states.dtype, "fc2_bias")
        _res_ok(residual, M * d, hidden_states.dtype)
        if fc1_blocked is None or fc2_blocked is None or not lib.mio_fused_mlp_blocked_weight_ok(M, d, I, act):
            raise ValueError("a blocked activation operand needs blocked weights and fused_mlp_blocked_weight_ok()")
        _blocked_sizes_ok(fc1_blocked, fc2_blocked, hidden_states, d, I, act)
        out = torch.empty(*x_blocked_shape, dtype=hidden_states.dtype, device=hidden_states.device)
        work = torch.empty((M + 255) // 256 * 256, I, dtype=hidden_states.dtype, device=hidden_states.device)
        r2 = None if residual is None else residual.reshape(-1, d)
        if r2 is not None and not r2.is_contiguous():
            r2 = r2.contiguous()
        if act == _lib.ACT_SWIGLU:
            _vec_ok(fc1_gate_bias, I, hidden_states.dtype, "fc1_gate_bias")
            check(lib.mio_fused_mlp_glu_fwd_bw(hidden_states.data_ptr(), fc1_blocked.data_ptr(), _ptr(fc1_bias),
                                               _ptr(fc1_gate_bias), fc2_blocked.data_ptr(), _ptr(fc2_bias), _ptr(r2),
                                               out.data_ptr(), work.data_ptr(), M, d, I, _dtype_id(hidden_states), 1, _stream()))
            return out
        check(lib.mio_fused_mlp_fwd_bw(hidden_states.data_ptr(), fc1_blocked.data_ptr(), _ptr(fc1_bias),
                                       fc2_blocked.data_ptr(), _ptr(fc2_bias), _ptr(r2), out.data_ptr(), work.data_ptr(),
                                       M, d, I, act, _dtype_id(hidden_states), 1, _stream()))
        return out
    if act == _lib.ACT_SWIGLU and fc1_gate_weight is None:
        raise ValueError("SwiGLU activation requires gate weights")
    dt = _dtype_id(hidden_states)
    d = hidden_states.shape[-1]
    I = fc1_weight.shape[0]
    if fc1_weight.shape[1] != d or tuple(fc2_weight.shape) != (d, I):
        raise ValueError("fc1/fc2 weight shapes do not match hidden size")
    x2 = _rows16(hidden_states.reshape(-1, d))
    if x2.stride(0) != d:
        x2 = x2.contiguous()
    M = x2.shape[0]
    _vec_ok(fc1_bias, I, hidden_states.dtype, "fc1_bias")
    _vec_ok(fc2_bias, d, hidden_states.dtype, "fc2_bias")
    if act == _lib.ACT_SWIGLU:
        _vec_ok(fc1_gate_bias, I, hidden_states.dtype, "fc1_gate_bias")
    _res_ok(residual, M * d, hidden_states.dtype)
    ws = [fc1_weight, fc2_weight] + ([fc1_gate_weight] if act == _lib.ACT_SWIGLU else [])
    for w_ in ws:
        if w_.dtype != hidden_states.dtype or not w_.is_contiguous():
            raise ValueError("weights must be contiguous and of the input dtype")
    out = torch.empty_like(hidden_states, memory_format=torch.contiguous_format)
    # workspace in whole 256-row blocks (mio_fused_mlp_workspace_bytes): the intermediate may use a blocked layout
    work = torch.empty((M + 255) // 256 * 256, I, dtype=hidden_states.dtype, device=hidden_states.device)
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, d)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    gate_w = fc1_gate_weight if act == _lib.ACT_SWIGLU else None
    gate_b = fc1_gate_bias if act == _lib.ACT_SWIGLU else None
    if fc1_blocked is not None and fc2_blocked is not None and lib.mio_fused_mlp_blocked_weight_ok(M, d, I, act):
        _blocked_sizes_ok(fc1_blocked, fc2_blocked, hidden_states, d, I, act)
        if act == _lib.ACT_SWIGLU:
            check(lib.mio_fused_mlp_glu_fwd_bw(x2.data_ptr(), fc1_blocked.data_ptr(), _ptr(fc1_bias), _ptr(gate_b),
                                               fc2_blocked.data_ptr(), _ptr(fc2_bias), _ptr(r2), out.data_ptr(),
                                               work.data_ptr(), M, d, I, dt, 0, _stream()))
            return out
        check(lib.mio_fused_mlp_fwd_bw(x2.data_ptr(), fc1_blocked.data_ptr(), _ptr(fc1_bias), fc2_blocked.data_ptr(),
                                       _ptr(fc2_bias), _ptr(r2), out.data_ptr(), work.data_ptr(), M, d, I, act, dt,
                                       0, _stream()))
        return out
    check(lib.mio_fused_mlp_fwd(x2.data_ptr(), fc1_weight.data_ptr(), _ptr(fc1_bias), _ptr(gate_w), _ptr(gate_b),
                                fc2_weight.data_ptr(), _ptr(fc2_bias), _ptr(r2), out.data_ptr(), work.data_ptr(),
                                M, d, I, act, dt, _stream()))
    return out


def layernorm(x, weight, bias=None, eps: float = 1e-5, residual=None, residual_alpha: float = 1.0,
              return_sum: bool = False, out_blocked: bool = False):
    """Drop-in for triton_layernorm (layernorm_kernels.py:191-276): optional x + alpha*residual first.
    out_blocked: y is returned as a [ceil(rows/256)*256, cols] tensor in the blocked activation layout
    (include/mio_hip.h) for a following gemm_bias_act / fused_mlp with x_blocked_shape=x.shape."""
    _need_cuda(x, weight)
    dt = _dtype_id(x)
    cols = x.shape[-1]
    _vec_ok(weight, cols, x.dtype, "layernorm weight")
    _vec_ok(bias, cols, x.dtype, "layernorm bias")
    _res_ok(residual, x.numel(), x.dtype)
    x2 = x.reshape(-1, cols)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    r2 = None
    if residual is not None:
        r2 = residual.reshape(-1, cols)
        if not r2.is_contiguous():
            r2 = r2.contiguous()
    s = torch.empty_like(x2) if (return_sum and r2 is not None) else None
    if out_blocked:
        if cols % 32 != 0:
            raise ValueError("out_blocked needs cols % 32 == 0")
        y = torch.empty((x2.shape[0] + 255) // 256 * 256, cols, dtype=x.dtype, device=x.device)
        check(lib.mio_layernorm_fwd_bx(x2.data_ptr(), _ptr(r2), weight.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(s),
                                       x2.shape[0], cols, float(eps), float(residual_alpha), dt, _stream()))
    else:
        y = torch.empty_like(x2)
        check(lib.mio_layernorm_fwd(x2.data_ptr(), _ptr(r2), weight.data_ptr(), _ptr(bias), y.data_ptr(), _ptr(s),
                                    x2.shape[0], cols, float(eps), float(residual_alpha), dt, _stream()))
        y = y.view(x.shape)
    if return_sum:
        return y, (s.view(x.shape) if s is not None else x)
    return y


# ------------------------------------------------------------------------------------------------
# paged decode
# ------------------------------------------------------------------------------------------------
def paged_attention_forward(query, output, k_cache, v_cache, block_tables, context_lengths, block_size: int,
                            max_seq_len: int, layer_idx: int, scale: Optional[float] = None) -> torch.Tensor:
    """Drop-in for triton_paged_attention_forward (attention_kernels.py:1206-1311).
    query/output [B,H,q_len,D] (output caller-preallocated, :1208,1286); caches
    [num_blocks, L, block_size, Hkv, D]; block_tables [B,max_blocks] int32; context_lengths [B] int32."""
    _need_cuda(query, output, k_cache, v_cache, block_tables, context_lengths)
    if query.dim() != 4 or output.shape != query.shape:
        raise ValueError("query/output must be [B,H,q_len,D] with equal shapes")
    if k_cache.dim() != 5 or v_cache.shape != k_cache.shape:
        raise ValueError("caches must be [num_blocks, num_layers, block_size, num_kv_heads, head_dim]")
    dt = _dtype_id(query)
    if k_cache.dtype != query.dtype or v_cache.dtype != query.dtype or output.dtype != query.dtype:
        raise ValueError("query, output and caches must share a dtype")
    if not (k_cache.is_contiguous() and v_cache.is_contiguous()):
        raise ValueError("caches must be contiguous")
    B, H, q_len, D = query.shape
    nb, L, bs, Hkv, Dc = k_cache.shape
    if bs != block_size or Dc != D:
        raise ValueError("cache geometry does not match block_size/head_dim")
    q = _rows16(query)
    if output.stride(-1) != 1:
        raise ValueError("output last dim must be contiguous")
    bt = block_tables.to(torch.int32).contiguous()
    cl = context_lengths.to(torch.int32).contiguous()
    sc = (1.0 / math.sqrt(D)) if scale is None else float(scale)
    nbytes = lib.mio_fa3_decode_workspace_bytes(B, H, q_len, D, int(max_seq_len))
    work = torch.empty(nbytes, dtype=torch.uint8, device=query.device)
    qs = (C.c_int64 * 3)(q.stride(0), q.stride(1), q.stride(2))
    os_ = (C.c_int64 * 3)(output.stride(0), output.stride(1), output.stride(2))
    check(lib.mio_fa3_decode_paged(q.data_ptr(), output.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                   bt.data_ptr(), cl.data_ptr(), qs, os_, B, H, Hkv, q_len, D, L, int(layer_idx), bs,
                                   bt.shape[1], int(max_seq_len), sc, dt, work.data_ptr(), _stream()))
    return output


def reshape_and_cache(key, value, k_cache, v_cache, block_tables, context_lengths, block_size: int, layer_idx: int):
    """Drop-in for triton_reshape_and_cache (attention_kernels.py:1314-1407): key/value [B,1,Hkv,D]."""
    _need_cuda(key, value, k_cache, v_cache)
    if key.dim() != 4 or key.shape[1] != 1:
        raise ValueError("reshape_and_cache supports q_seq_len == 1 only (attention_kernels.py:1363-1365)")
    dt = _dtype_id(key)
    B, _, Hkv, D = key.shape
    nb, L, bs, Hc, Dc = k_cache.shape
    if (Hc, Dc, bs) != (Hkv, D, block_size):
        raise ValueError("cache geometry mismatch")
    key, value = _rows16(key), _rows16(value)
    bt = block_tables.to(torch.int32).contiguous()
    cl = context_lengths.to(torch.int32).contiguous()
    ks = (C.c_int64 * 2)(key.stride(0), key.stride(2))
    vs = (C.c_int64 * 2)(value.stride(0), value.stride(2))
    check(lib.mio_reshape_and_cache(key.data_ptr(), value.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                    bt.data_ptr(), cl.data_ptr(), ks, vs, B, Hkv, D, L, int(layer_idx), bs,
                                    bt.shape[1], dt, _stream()))


# ------------------------------------------------------------------------------------------------
# composed entry points: QKV projection + attention + out-projection, LayerNorm + QKV projection
# ------------------------------------------------------------------------------------------------
def fused_attention(hidden_states: torch.Tensor, qkv_weight: torch.Tensor, qkv_bias: Optional[torch.Tensor],
                    out_weight: torch.Tensor, out_bias: Optional[torch.Tensor], mask: Optional[torch.Tensor] = None,
                    causal: bool = False, num_heads: int = 8, head_dim: Optional[int] = None, dropout_p: float = 0.0,
                    softmax_scale: Optional[float] = None, block_size: int = 128) -> torch.Tensor:
    """Drop-in for triton_fused_attention (flash_attention_kernels.py:1361-1530): hidden [B,S,d],
    qkv_weight [3d,d], out_weight [d,d].  Three launches (QKV GEMM, tiled attention reading q/k/v as strided views
    of the QKV result, out-projection GEMM); nothing is copied in between."""
    if hidden_states.dim() != 3:
        raise ValueError(f"Expected 3D input tensor, got shape: {hidden_states.shape}")
    B, S, d = hidden_states.shape
    D = head_dim if head_dim is not None else d // num_heads
    if num_heads * D != d or qkv_weight.shape[0] != 3 * d:
        raise ValueError(f"hidden_size {d} does not match num_heads {num_heads} x head_dim {D} / qkv_weight "
                         f"{tuple(qkv_weight.shape)}")
    qkv = gemm_bias_act(hidden_states, qkv_weight, qkv_bias).view(B, S, 3, num_heads, D)
    ctx = flash_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], mask=mask, causal=causal,
                          softmax_scale=softmax_scale, dropout_p=dropout_p, block_size=block_size)
    return gemm_bias_act(ctx.view(B, S, d), out_weight, out_bias)


def _infer_heads(hidden_size: int, num_heads: int) -> int:
    if num_heads:
        return num_heads
    for hs in (64, 80, 128):  # fused_layernorm_qkv.py:653-663
        if hidden_size % hs == 0:
            return hidden_size // hs
    return max(1, hidden_size // 64)


def fused_layernorm_qkv(hidden_states, layernorm_weight, layernorm_bias, query_weight, key_weight, value_weight,
                        query_bias=None, key_bias=None, value_bias=None, eps: float = 1e-5, num_heads: int = 0,
                        num_kv_heads: Optional[int] = None):
    """Drop-in for triton_fused_layernorm_qkv / pytorch_fused_layernorm_qkv (fused_layernorm_qkv.py:422-700):
    returns (q [B,S,H,Dh], k [B,S,Hkv,Dkv], v [B,S,Hkv,Dkv]).

    Two kernel kinds: the row-wise LayerNorm (HBM-bound, ~3 % of a layer at the benchmark shape) and the MFMA GEMM
    whose operand tiles are moved global -> LDS by DMA; normalising inside that GEMM would put every activation
    tile through registers and the vector ALU, which costs the matrix pipeline more than the one [B,S,d] round
    trip it saves."""
    if hidden_states.dim() != 3:
        raise ValueError(f"Expected 3D input tensor, got shape: {hidden_states.shape}")
    B, S, d = hidden_states.shape
    H = _infer_heads(d, num_heads)
    Hkv = H if num_kv_heads is None else num_kv_heads
    xn = layernorm(hidden_states, layernorm_weight, layernorm_bias, eps)
    q = gemm_bias_act(xn, query_weight, query_bias)
    k = gemm_bias_act(xn, key_weight, key_bias)
    v = gemm_bias_act(xn, value_weight, value_bias)
    return (q.view(B, S, H, q.shape[-1] // H), k.view(B, S, Hkv, k.shape[-1] // Hkv),
            v.view(B, S, Hkv, v.shape[-1] // Hkv))


def flash_compatible_wrapper(hidden_states, layernorm_weight, layernorm_bias, qkv_weight, qkv_bias=None,
                             eps: float = 1e-5, num_heads: int = 0, num_kv_heads: Optional[int] = None):
    """fused_layernorm_qkv.py:1073-1116: combined [3d,d] QKV weight -> ONE GEMM; q/k/v are strided views of it."""
    if hidden_states.dim() != 3:
        raise ValueError(f"Expected 3D input tensor, got shape: {hidden_states.shape}")
    B, S, d = hidden_states.shape
    H = _infer_heads(d, num_heads)
    Hkv = H if num_kv_heads is None else num_kv_heads
    xn = layernorm(hidden_states, layernorm_weight, layernorm_bias, eps)
    qkv = gemm_bias_act(xn, qkv_weight, qkv_bias)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    return q.view(B, S, H, d // H), k.view(B, S, Hkv, d // Hkv), v.view(B, S, Hkv, d // Hkv)


def ring_compatible_wrapper(hidden_states, layernorm_weight, layernorm_bias, q_weight, k_weight, v_weight,
                            q_bias=None, k_bias=None, v_bias=None, eps: float = 1e-5, num_heads: int = 0,
                            num_kv_heads: Optional[int] = None):
    """fused_layernorm_qkv.py:1118-1161: as fused_layernorm_qkv, head-major [B,H,S,Dh] views for ring attention."""
    q, k, v = fused_layernorm_qkv(hidden_states, layernorm_weight, layernorm_bias, q_weight, k_weight, v_weight,
                                  q_bias, k_bias, v_bias, eps, num_heads, num_kv_heads)
    return q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)


# reference-name aliases (drop-in for code written against the Triton wrappers)
triton_flash_attention = flash_attention
triton_ring_attention_forward = ring_attention_forward
triton_fused_mlp = fused_mlp
triton_layernorm = layernorm
triton_paged_attention_forward = paged_attention_forward
triton_reshape_and_cache = reshape_and_cache
triton_fused_attention = fused_attention
triton_fused_layernorm_qkv = fused_layernorm_qkv
