"""CPU oracle: attention (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py).

Every function states the reference lines it restates.  Math is done in fp64 or fp32 on the
host with torch; inputs may be fp32/bf16/fp16 and are up-cast first, so the oracle sees exactly
the values the HIP kernels see.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch

NEG_FILL = -1e9  # reference mask fill value (flash_attention_kernels.py:254,273; flash_attention.py:1224)


def _canon_keep_mask(mask: torch.Tensor) -> torch.Tensor:
    """Canonicalise a keep-mask (1 = attend, 0 = masked) to [B, 1|H, 1|Sq, Sk].

    Restates flash_attention_kernels.py:1232-1257 (2-D [B,S] -> [B,1,1,S];
    3-D [B,1,S] -> [B,1,1,S]; 3-D [B,S,S] -> [B,1,S,S]; 4-D kept)."""
    if mask.dim() == 2:
        return mask[:, None, None, :]
    if mask.dim() == 3:
        if mask.shape[1] == 1:
            return mask[:, :, None, :]
        return mask[:, None, :, :]
    if mask.dim() == 4:
        return mask
    raise ValueError(f"Unsupported mask shape: {tuple(mask.shape)}")


def standard_attention(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
    causal: bool = False,
    softmax_scale: Optional[float] = None,
    additive_mask: Optional[torch.Tensor] = None,
    q_offset: int = 0,
    k_offset: int = 0,
    dtype=torch.float64,
) -> torch.Tensor:
    """Dense exact-softmax attention, layout [B, S, H, D] -> [B, Sq, H, D].

    Restates the reference's own comparator `standard_attention`
    (kernels/attention/flash_attention.py:1216-1229): einsum("bshd,bkhd->bhsk") * scale,
    masked_fill(triu(diagonal=1), -1e9), softmax(-1), einsum("bhsk,bkhd->bshd"); the keep-mask
    rule `scores*mask + (-1e9)*(1-mask)` is flash_attention_kernels.py:254,273.
    GQA (Hkv < H) follows flash_attention.py:894-912 (repeat_interleave of K/V heads).
    `q_offset`/`k_offset` shift absolute positions for the causal rule (ring shards).
    """
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    qf, kf, vf = q.to(dtype), k.to(dtype), v.to(dtype)
    if Hkv != H:
        rep = H // Hkv
        kf = kf.repeat_interleave(rep, dim=2)
        vf = vf.repeat_interleave(rep, dim=2)
    s = torch.einsum("bshd,bkhd->bhsk", qf, kf) * scale
    if causal:
        qi = torch.arange(Sq)[:, None] + q_offset
        ki = torch.arange(Sk)[None, :] + k_offset
        s = s.masked_fill((ki > qi)[None, None], NEG_FILL)
    if mask is not None:
        m = _canon_keep_mask(mask).to(dtype)
        s = s * m + NEG_FILL * (1.0 - m)
    if additive_mask is not None:
        s = s + additive_mask.to(dtype)
    p = torch.softmax(s, dim=-1)
    return torch.einsum("bhsk,bkhd->bshd", p, vf)


def attention_with_lse(
    q, k, v, mask=None, causal=False, softmax_scale=None, additive_mask=None,
    q_offset: int = 0, k_offset: int = 0, dtype=torch.float64,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same as `standard_attention` but also returns lse [B, H, Sq] (natural log of the softmax
    denominator, i.e. m + log(l) of flash_attention_kernels.py:276-305's running stats).

    Unlike the -1e9 fill, causal positions with offsets are treated as truly absent
    (score = -inf): a ring shard whose keys all lie in the future contributes nothing
    (lse = -inf, o = 0).  With q_offset == k_offset == 0 both conventions agree because the
    diagonal is always present.
    """
    B, Sq, H, D = q.shape
    Sk, Hkv = k.shape[1], k.shape[2]
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    qf, kf, vf = q.to(dtype), k.to(dtype), v.to(dtype)
    if Hkv != H:
        rep = H // Hkv
        kf = kf.repeat_interleave(rep, dim=2)
        vf = vf.repeat_interleave(rep, dim=2)
    s = torch.einsum("bshd,bkhd->bhsk", qf, kf) * scale
    if causal:
        qi = torch.arange(Sq)[:, None] + q_offset
        ki = torch.arange(Sk)[None, :] + k_offset
        s = s.masked_fill((ki > qi)[None, None], float("-inf"))
    if mask is not None:
        m = _canon_keep_mask(mask).to(dtype)
        s = torch.where(m != 0, s, torch.full_like(s, NEG_FILL))
    if additive_mask is not None:
        s = s + additive_mask.to(dtype)
    lse = torch.logsumexp(s, dim=-1)
    p = torch.exp(s - torch.where(torch.isinf(lse), torch.zeros_like(lse), lse)[..., None])
    p = torch.where(torch.isinf(lse)[..., None], torch.zeros_like(p), p)
    o = torch.einsum("bhsk,bkhd->bshd", p, vf)
    return o, lse


def flash_attention_online(
    q: torch.Tensor,
    k: torch.Tensor,
    v: torch.Tensor,
    mask: Optional[torch.Tensor] = None,
    causal: bool = False,
    softmax_scale: Optional[float] = None,
    block_size: int = 128,
    dtype=torch.float32,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Block-wise online-softmax restatement of `_flash_attention_forward_kernel`
    (kernels/triton/flash_attention_kernels.py:176-305), layout [B, S, H, D].

    Per KV block: scores = (q @ k^T) * scale (:242-244); causal `row >= col` fill -1e9
    (:247-254); keep-mask fill -1e9 (:257-273); m_block = max, m_new = max(m_i, m_block),
    alpha = exp(m_i - m_new), beta = exp(m_block - m_new) (:278-288);
    l = alpha*l + beta*sum(exp(s - m_block)) (:290-292); o = alpha*o + exp(s - m_new) @ v
    (:294-298); final o / l (:305).  Returns (o [B,S,H,D], l [B,H,S], m [B,H,S]) like the
    kernel's STORE_L_M outputs (:308-325).
    """
    B, Sq, H, D = q.shape
    Sk = k.shape[1]
    scale = (1.0 / math.sqrt(D)) if softmax_scale is None else softmax_scale
    qf = q.to(dtype).permute(0, 2, 1, 3)  # [B,H,S,D]
    kf = k.to(dtype).permute(0, 2, 1, 3)
    vf = v.to(dtype).permute(0, 2, 1, 3)
    if kf.shape[1] != H:
        rep = H // kf.shape[1]
        kf = kf.repeat_interleave(rep, dim=1)
        vf = vf.repeat_interleave(rep, dim=1)
    keep = None if mask is None else _canon_keep_mask(mask).to(dtype)
    o = torch.zeros(B, H, Sq, D, dtype=dtype)
    m_i = torch.full((B, H, Sq), float("-inf"), dtype=dtype)
    l_i = torch.zeros(B, H, Sq, dtype=dtype)
    rows = torch.arange(Sq)[:, None]
    for start in range(0, Sk, block_size):
        end = min(start + block_size, Sk)
        s = torch.matmul(qf, kf[:, :, start:end].transpose(-1, -2)) * scale
        if causal:
            cols = torch.arange(start, end)[None, :]
            cm = (rows >= cols).to(dtype)
            s = s * cm + NEG_FILL * (1.0 - cm)
        if keep is not None:
            mb = keep[..., start:end]
            s = s * mb + NEG_FILL * (1.0 - mb)
        m_block = s.max(dim=-1).values
        m_new = torch.maximum(m_i, m_block)
        alpha = torch.exp(m_i - m_new)
        beta = torch.exp(m_block - m_new)
        l_i = alpha * l_i + beta * torch.exp(s - m_block[..., None]).sum(-1)
        o = alpha[..., None] * o + torch.matmul(torch.exp(s - m_new[..., None]), vf[:, :, start:end])
        m_i = m_new
    o = o / l_i[..., None]
    return o.permute(0, 2, 1, 3).contiguous(), l_i, m_i


def ring_attention_forward(
    query: torch.Tensor,
    key: torch.Tensor,
    value: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    chunk_size: int = 128,
    dtype=torch.float32,
) -> torch.Tensor:
    """Ring attention, semantics (A): exact attention via online softmax over KV chunks.

    Restates the PyTorch ring fallback `triton_ring_attention_forward`
    (kernels/triton/attention_kernels.py:1520-1591) which mirrors
    `_ring_attention_forward_kernel` (:164-193): query pre-scaled by 1/sqrt(D) (:1532),
    chunk = min(128, Sk) (:1535), additive mask [B,1|H,Sq,Sk] (:1565-1566),
    running (m, l, acc) (:1568-1585), final /l (:1588).
    Layout: q,k,v [B, H, S, D] (head-major) -> [B, Sq, H*D].
    """
    B, H, Sq, D = query.shape
    Sk = key.shape[2]
    qf = query.to(dtype) * (1.0 / math.sqrt(D))
    kf, vf = key.to(dtype), value.to(dtype)
    chunk = min(chunk_size, Sk)
    out = torch.zeros(B, H, Sq, D, dtype=dtype)
    m_i = torch.full((B, H, Sq, 1), float("-inf"), dtype=dtype)
    l_i = torch.zeros(B, H, Sq, 1, dtype=dtype)
    for k0 in range(0, Sk, chunk):
        k1 = min(k0 + chunk, Sk)
        w = torch.matmul(qf, kf[:, :, k0:k1].transpose(-1, -2))
        if attention_mask is not None:
            w = w + attention_mask[:, :, :, k0:k1].to(dtype)
        m_ij = w.max(dim=-1, keepdim=True).values
        p_ij = torch.exp(w - m_ij)
        l_ij = p_ij.sum(dim=-1, keepdim=True)
        m_new = torch.maximum(m_i, m_ij)
        alpha = torch.exp(m_i - m_new)
        beta = torch.exp(m_ij - m_new)
        out = out * alpha
        l_i = l_i * alpha + beta * l_ij
        m_i = m_new
        out = out + torch.matmul(p_ij, vf[:, :, k0:k1]) * beta
    out = out / l_i
    return out.permute(0, 2, 1, 3).contiguous().view(B, Sq, H * D)


def merge_attention_states(
    o_a: torch.Tensor, lse_a: torch.Tensor, o_b: torch.Tensor, lse_b: torch.Tensor
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Merge two normalised partial attention results over disjoint key sets.

    o_* [B, S, H, D], lse_* [B, H, S].  This is the (alpha, beta) update of
    attention_kernels.py:1573-1585 written on normalised states: with w_x = exp(lse_x - lse),
    lse = logaddexp(lse_a, lse_b), o = w_a*o_a + w_b*o_b.  A state with lse = -inf is empty.
    """
    dt = torch.float64
    la, lb = lse_a.to(dt), lse_b.to(dt)
    lse = torch.logaddexp(la, lb)
    safe = torch.where(torch.isinf(lse) & (lse < 0), torch.zeros_like(lse), lse)
    wa = torch.exp(la - safe).permute(0, 2, 1)[..., None]
    wb = torch.exp(lb - safe).permute(0, 2, 1)[..., None]
    o = wa * o_a.to(dt) + wb * o_b.to(dt)
    return o, lse


def paged_attention_forward(
    query: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    context_lengths: torch.Tensor,
    block_size: int,
    layer_idx: int,
    scale: Optional[float] = None,
    dtype=torch.float64,
) -> torch.Tensor:
    """Attention of q [B, H, q_len, D] over a paged KV cache, restating
    `_paged_attention_fwd_kernel` (kernels/triton/attention_kernels.py:628-808).

    Cache layout [num_blocks, num_layers, block_size, Hkv, D] (:645); token t of sequence b
    lives in physical block block_tables[b, t // block_size] at slot t % block_size (:728-751);
    keys at positions >= context_lengths[b] are excluded (:771-777); there is NO causal mask
    inside the kernel (the causal line is commented out, :774-776); rows with no keys give 0
    (`l_i == 0 -> 1`, :802).  Returns [B, H, q_len, D].
    """
    B, H, q_len, D = query.shape
    Hkv = k_cache.shape[3]
    sc = (1.0 / math.sqrt(D)) if scale is None else scale
    out = torch.zeros(B, H, q_len, D, dtype=dtype)
    for b in range(B):
        n = int(context_lengths[b])
        if n == 0:
            continue
        pos = torch.arange(n)
        blk = block_tables[b, pos // block_size].long()
        slot = pos % block_size
        kk = k_cache[blk, layer_idx, slot].to(dtype)  # [n, Hkv, D]
        vv = v_cache[blk, layer_idx, slot].to(dtype)
        if Hkv != H:
            kk = kk.repeat_interleave(H // Hkv, dim=1)
            vv = vv.repeat_interleave(H // Hkv, dim=1)
        s = torch.einsum("hqd,nhd->hqn", query[b].to(dtype), kk) * sc
        p = torch.softmax(s, dim=-1)
        out[b] = torch.einsum("hqn,nhd->hqd", p, vv)
    return out


def reshape_and_cache(
    key: torch.Tensor,
    value: torch.Tensor,
    k_cache: torch.Tensor,
    v_cache: torch.Tensor,
    block_tables: torch.Tensor,
    context_lengths: torch.Tensor,
    block_size: int,
    layer_idx: int,
) -> None:
    """Scatter the current token's K/V into the paged cache (in place), restating
    `_reshape_and_cache_kernel` (kernels/triton/attention_kernels.py:811-905): for sequence b
    the token position is context_lengths[b] - 1 (:858), physical block
    block_tables[b, pos // block_size], slot pos % block_size (:861-870).
    key/value: [B, 1, Hkv, D] (q_len == 1 only, :1363-1365)."""
    B = key.shape[0]
    for b in range(B):
        pos = int(context_lengths[b]) - 1
        if pos < 0:
            continue
        blk = int(block_tables[b, pos // block_size])
        slot = pos % block_size
        k_cache[blk, layer_idx, slot] = key[b, 0].to(k_cache.dtype)
        v_cache[blk, layer_idx, slot] = value[b, 0].to(v_cache.dtype)
