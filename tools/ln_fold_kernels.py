#!/usr/bin/env python3
"""Per-kernel cost of the LayerNorm fold at the C2 layer shapes (M 32768, d 1024, I 4096): each folded GEMM (ops.gemm_ln) beside the
launch it replaces, HIP events over back-to-back launches (bench.py _events_ms)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
    sys.path.insert(0, p)
from mio import ops
from bench import _events_ms

dt = torch.bfloat16
M, d, I = 32768, 1024, 4096
torch.manual_seed(0)
x = torch.randn(M, d, device="cuda", dtype=dt)
g, b = (1 + 0.1 * torch.randn(d, device="cuda")).to(dt), (0.1 * torch.randn(d, device="cuda")).to(dt)
wqkv, bqkv = (torch.randn(3 * d, d, device="cuda") * 0.02).to(dt), (torch.randn(3 * d, device="cuda") * 0.02).to(dt)
wo, bo = (torch.randn(d, d, device="cuda") * 0.02).to(dt), (torch.randn(d, device="cuda") * 0.02).to(dt)
w1, b1 = (torch.randn(I, d, device="cuda") * 0.02).to(dt), (torch.randn(I, device="cuda") * 0.02).to(dt)
w2, b2 = (torch.randn(d, I, device="cuda") * 0.02).to(dt), (torch.randn(d, device="cuda") * 0.02).to(dt)
wqkv_b, wo_b, w1_b, w2_b = (ops.block_weight(t) for t in (wqkv, wo, w1, w2))
cs = (d, 2 * d, 0.18)
# the stream as a producer leaves it
xb, st = ops.gemm_ln(x, wo_b, bo, M=M, N=d, K=d, residual=x, out_blocked=True, stats_out=True)
lnb = ops.layernorm(x.view(8, 4096, d), g, b, out_blocked=True)
fq = ops.ln_fold_weight(wqkv, g, b, bqkv)
f1 = ops.ln_fold_weight(w1, g, b, b1)
hb, _ = ops.gemm_ln(xb, f1[0], f1[1], M=M, N=I, K=d, activation="gelu", x_blocked=True, out_blocked=True, ln_stats=st)
o3 = torch.empty(8, 4096, 3 * d, device="cuda", dtype=dt)
rows = [
    ("layernorm (blocked out)", lambda: ops.layernorm(x.view(8, 4096, d), g, b, out_blocked=True)),
    ("qkv   separate", lambda: ops.gemm_bias_act(lnb, wqkv, bqkv, out=o3, w_blocked=wqkv_b, col_scale=cs, x_blocked_shape=(8, 4096, d))),
    ("qkv   folded  ", lambda: ops.gemm_ln(xb, fq[0], fq[1], M=M, N=3 * d, K=d, x_blocked=True, ln_stats=st, col_scale=cs)),
    ("oproj separate", lambda: ops.gemm_bias_act(lnb, wo, bo, residual=x.view(8, 4096, d), w_blocked=wo_b, x_blocked_shape=(8, 4096, d))),
    ("oproj producer", lambda: ops.gemm_ln(lnb, wo_b, bo, M=M, N=d, K=d, x_blocked=True, residual=xb, res_blocked=True, out_blocked=True, stats_out=True)),
    ("oproj blocked res/out, no stats", lambda: ops.gemm_ln(lnb, wo_b, bo, M=M, N=d, K=d, x_blocked=True, residual=xb, res_blocked=True, out_blocked=True)),
    ("fc1   separate", lambda: ops.gemm_ln(lnb, w1_b, b1, M=M, N=I, K=d, activation="gelu", x_blocked=True, out_blocked=True)),
    ("fc1   folded  ", lambda: ops.gemm_ln(xb, f1[0], f1[1], M=M, N=I, K=d, activation="gelu", x_blocked=True, out_blocked=True, ln_stats=st)),
    ("fc2   separate", lambda: ops.gemm_ln(hb, w2_b, b2, M=M, N=d, K=I, x_blocked=True, residual=x)),
    ("fc2   producer", lambda: ops.gemm_ln(hb, w2_b, b2, M=M, N=d, K=I, x_blocked=True, residual=xb, res_blocked=True, out_blocked=True, stats_out=True)),
]
for name, fn in rows:
    ms = sorted(_events_ms(fn, 20) for _ in range(3))[1]
    print(f"{name:36s} {ms * 1e3:8.1f} us", flush=True)
