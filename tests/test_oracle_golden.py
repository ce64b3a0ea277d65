"""The CPU oracle against the golden vectors generated from the reference's runnable code
(tests/golden/make_golden.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle


def _load(golden_dir, name):
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(golden_dir, name)).items()}


@pytest.mark.parametrize("name,act", [("gelu", "gelu"), ("swiglu", "swiglu"), ("relu", "relu"),
                                      ("silu", "silu"), ("gelu_erf", "gelu_erf")])
def test_mlp_modules(golden_dir, name, act):
    g = _load(golden_dir, "fused_mlp_modules.npz")
    pre = f"{name}_mlp_" if f"{name}_mlp_fc1_weight" in g else f"{name}_"
    y = oracle.fused_mlp(g[f"{name}_x"], g[pre + "fc1_weight"], g[pre + "fc1_bias"], g[pre + "fc2_weight"],
                         g[pre + "fc2_bias"], act, g.get(pre + "fc1_gate_weight"), g.get(pre + "fc1_gate_bias"))
    assert (y - g[f"{name}_y"].double()).abs().max() < 2e-6


@pytest.mark.parametrize("act,oact", [("gelu", "gelu_erf"), ("relu", "relu"), ("swiglu", "swiglu")])
def test_mlp_functional(golden_dir, act, oact):
    g = _load(golden_dir, "fused_mlp_functional.npz")
    y = oracle.fused_mlp(g[f"{act}_x"], g[f"{act}_w1"], g[f"{act}_b1"], g[f"{act}_w2"], g[f"{act}_b2"], oact,
                         g[f"{act}_wg"], g[f"{act}_bg"])
    assert (y - g[f"{act}_y"].double()).abs().max() < 5e-6


RING_CASES = ["d64_nomask", "d64_additive", "d64_causal", "d80_cross", "d128_causal", "d64_padding"]


@pytest.mark.parametrize("case", RING_CASES)
def test_ring_fallback(golden_dir, case):
    g = _load(golden_dir, "ring_attention_fallback.npz")
    q, k, v, o = g[f"{case}_q"], g[f"{case}_k"], g[f"{case}_v"], g[f"{case}_o"]
    B, H, Sq, D = q.shape
    Sk = k.shape[2]
    mask = None
    if f"{case}_mask" in g:
        mask = g[f"{case}_mask"]
    elif "causal" in case:
        mask = torch.triu(torch.full((Sq, Sk), -1e9), diagonal=1)[None, None].expand(B, 1, Sq, Sk)
    elif "padding" in case:
        mask = ((1.0 - g[f"{case}_keep"]) * -1e9)[:, None, None, :].expand(B, 1, Sq, Sk)
    # (1) chunked restatement == the reference fallback
    o1 = oracle.ring_attention_forward(q, k, v, mask)
    assert (o1 - o).abs().max() < 2e-6
    # (2) the dense exact-softmax oracle agrees with it (this is what the HIP kernels are held to)
    qs, ks, vs = (t.permute(0, 2, 1, 3) for t in (q, k, v))
    kw = {}
    if "causal" in case:
        kw["causal"] = True
    elif "padding" in case:
        kw["mask"] = g[f"{case}_keep"]
    elif mask is not None:
        kw["additive_mask"] = mask
    o2 = oracle.standard_attention(qs, ks, vs, **kw).reshape(B, Sq, H * D)
    assert (o2 - o.double()).abs().max() < 3e-6
    o3, lse = oracle.attention_with_lse(qs, ks, vs, **kw)
    assert (o3.reshape(B, Sq, H * D) - o.double()).abs().max() < 3e-6
    # (3) the online-softmax restatement of the FA3 kernel math agrees as well
    if "additive" not in case:
        okw = {"causal": "causal" in case}
        if "padding" in case:
            okw["mask"] = g[f"{case}_keep"]
        o4, l, m = oracle.flash_attention_online(qs, ks, vs, block_size=64, **okw)
        assert (o4.reshape(B, Sq, H * D) - o).abs().max() < 3e-6
        assert ((m + l.log()).double() - lse).abs().max() < 1e-4


def test_block_size_invariance_and_merge():
    torch.manual_seed(0)
    q, k, v = (torch.randn(1, 70, 2, 32) for _ in range(3))
    ref = oracle.standard_attention(q, k, v, causal=True)
    for bs in (16, 33, 128):
        o, _, _ = oracle.flash_attention_online(q, k, v, causal=True, block_size=bs, dtype=torch.float64)
        assert (o - ref).abs().max() < 1e-12
    # split-KV merge reproduces the whole
    oa, la = oracle.attention_with_lse(q, k[:, :30], v[:, :30])
    ob, lb = oracle.attention_with_lse(q, k[:, 30:], v[:, 30:])
    o, lse = oracle.merge_attention_states(oa, la, ob, lb)
    full, lfull = oracle.attention_with_lse(q, k, v)
    assert (o - full).abs().max() < 1e-12 and (lse - lfull).abs().max() < 1e-12
    # causal with offsets: a shard entirely in the future is empty
    oe, le = oracle.attention_with_lse(q, k, v, causal=True, q_offset=0, k_offset=1000)
    assert torch.isinf(le).all() and oe.abs().max() == 0
    o2, l2 = oracle.merge_attention_states(full, lfull, oe, le)
    assert (o2 - full).abs().max() < 1e-12


def test_layernorm(golden_dir):
    g = _load(golden_dir, "layernorm.npz")
    y = oracle.layernorm(g["x"], g["w"], g["b"], 1e-5)
    assert (y - g["y"].double()).abs().max() < 5e-6
    y2, _ = oracle.layernorm_residual(g["x"], g["r"], g["w"], g["b"], 1e-5, 0.5)
    assert (y2 - g["y_res"].double()).abs().max() < 5e-6


def test_paged_matches_dense():
    torch.manual_seed(1)
    B, H, D, bs, L = 2, 4, 32, 16, 2
    ctx = torch.tensor([37, 16], dtype=torch.int32)
    nblk = 8
    kc = torch.randn(nblk, L, bs, H, D)
    vc = torch.randn(nblk, L, bs, H, D)
    bt = torch.tensor([[5, 2, 7, 0], [3, 1, 0, 0]], dtype=torch.int32)
    q = torch.randn(B, H, 1, D)
    out = oracle.paged_attention_forward(q, kc, vc, bt, ctx, bs, 1)
    for b in range(B):
        n = int(ctx[b])
        ks = torch.cat([kc[bt[b, i], 1] for i in range((n + bs - 1) // bs)])[:n]
        vs = torch.cat([vc[bt[b, i], 1] for i in range((n + bs - 1) // bs)])[:n]
        ref = oracle.standard_attention(q[b].permute(1, 0, 2)[None], ks[None], vs[None])
        assert (out[b].permute(1, 0, 2)[None] - ref).abs().max() < 1e-10
    # reshape_and_cache writes the slot the next decode step reads
    knew, vnew = torch.randn(B, 1, H, D), torch.randn(B, 1, H, D)
    ctx2 = ctx + 1
    oracle.reshape_and_cache(knew, vnew, kc, vc, bt, ctx2, bs, 1)
    assert torch.equal(kc[bt[0, 37 // bs], 1, 37 % bs], knew[0, 0])
    assert torch.equal(vc[bt[1, 16 // bs], 1, 16 % bs], vnew[1, 0])


def test_baseline_runner_plumbing():
    from oracle.baseline_runner import time_cpu_baseline
    r = time_cpu_baseline(hidden_size=64, num_heads=4, num_layers=2, batch=1, seq_len=32, warmup=1, iters=1)
    assert r["tokens_per_s"] > 0
