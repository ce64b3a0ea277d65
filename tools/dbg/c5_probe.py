#!/usr/bin/env python3
"""Per-op timing of the C5 block (d 1280, H 16 -> Dh 80, Sq = Sk = 4096, B 8, I 5120)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops
from bench import _events_ms
B, S, d, H, I = 8, 4096, 1280, 16, 5120
D = d // H
dt = torch.bfloat16
M = B * S
x = torch.randn(B, S, d, device="cuda", dtype=dt)
def w(n, k): return (torch.randn(n, k, device="cuda") * 0.02).to(dt)
wq, w1, w2 = w(d, d), w(I, d), w(d, I)
b = torch.zeros(d, device="cuda", dtype=dt); b1 = torch.zeros(I, device="cuda", dtype=dt)
for name, fn, fl in [
    ("proj N1280 K1280", lambda: ops.gemm_bias_act(x, wq, b), 2.0 * M * d * d),
    ("proj blocked", (lambda wb=ops.block_weight(wq): ops.gemm_bias_act(x, wq, b, w_blocked=wb)), 2.0 * M * d * d),
    ("proj+res blocked", (lambda wb=ops.block_weight(wq): ops.gemm_bias_act(x, wq, b, residual=x, w_blocked=wb)), 2.0 * M * d * d),
    ("fused_mlp plain", lambda: ops.fused_mlp(x, w1, b1, w2, b, "gelu", residual=x), 4.0 * M * d * I),
    ("fused_mlp blocked", (lambda a=ops.block_weight(w1), c=ops.block_weight(w2): ops.fused_mlp(x, w1, b1, w2, b, "gelu", residual=x, fc1_blocked=a, fc2_blocked=c)), 4.0 * M * d * I),
    ("layernorm", lambda: ops.layernorm(x, b, b), 0),
]:
    t = _events_ms(fn, 10)
    print(f"{name}: {t:.3f} ms  {fl / t / 1e9:.0f} TFLOP/s", flush=True)
q = torch.randn(B, S, H, D, device="cuda", dtype=dt)
k = torch.randn(B, S, H, D, device="cuda", dtype=dt)
v = torch.randn(B, S, H, D, device="cuda", dtype=dt)
fl = 4.0 * B * S * S * d
t = _events_ms(lambda: ops.fa3_fwd(q, k, v), 10); print(f"fa3_fwd bshd D80: {t:.3f} ms {fl / t / 1e9:.0f} TFLOP/s", flush=True)
qh, kh, vh = (t_.permute(0, 2, 1, 3) for t_ in (q, k, v))
t = _events_ms(lambda: ops.ring_attention_forward(qh, kh, vh), 10); print(f"ring_attention_forward (head-major views) D80: {t:.3f} ms {fl / t / 1e9:.0f} TFLOP/s", flush=True)
from mio.synthetic import CrossBlock
blk = CrossBlock(d, H, I).to("cuda", dt).eval()
with torch.no_grad():
    t = _events_ms(lambda: blk(x, x), 5); print(f"CrossBlock: {t:.3f} ms", flush=True)
    t = _events_ms(lambda: blk.attn(x, x, residual=x), 5); print(f"  attn module: {t:.3f} ms", flush=True)
    t = _events_ms(lambda: blk.mlp(x, residual=x, pre_norm=blk.ln_2), 5); print(f"  mlp module: {t:.3f} ms", flush=True)
