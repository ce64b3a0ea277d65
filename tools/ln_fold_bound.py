#!/usr/bin/env python3
"""Measured bound for SURVEY 8(f)-2's LayerNorm -> GEMM fold (rank-1 epilogue form): same process, same box,
interleaved, ms per forward of the C2 stack
  A  as shipped: two LayerNorm launches per layer, blocked hand-over into QKV / fc1
  B  the fused form's BEST case: no LayerNorm launch at all, QKV / fc1 read the row-major residual stream directly (what
     they would do with the fold); the fold's own costs -- a row-statistics pass or finalize launches, one fma per output
     element in two read-outs -- are NOT charged.  (Values are not LayerNorm'd: timing only.)
  C  as B but with the statistics pass a fold needs if the row sums do not come out of the previous GEMM: one read of the
     residual stream per LayerNorm (here: the shipped kernel writing into a scratch buffer, an upper bound for that pass).
B - A is the most the fold can win; C - A what the simple (statistics-kernel) variant can win at best."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops, _nn
from mio.kernels.attention import flash_attention as fa_mod
from mio.kernels.mlp import fused_mlp as mlp_mod
from mio.synthetic import GPT2ShapedStack

B, S, d, H, L, I = 8, 4096, 1024, 16, 24, 4096
dt = torch.bfloat16
model = GPT2ShapedStack(d, H, L, I, causal=True, precision="bf16", seed=0).to("cuda", dt).eval()
model.no_ln_fold = True  # this tool measured the ROW-MAJOR fold bound before the fold was built: separate LayerNorm kernels as mode A
x = torch.randn(B, S, d, device="cuda", dtype=dt)
real_prenorm = _nn.prenorm_linear
real_ln = ops.layernorm
mode = {"m": "A"}
scratch = torch.empty(B, S, d, device="cuda", dtype=dt)


def prenorm(xx, ln, lin, cache, dtype, activation="none", residual=None, col_scale=None):
    if mode["m"] == "A":
        return real_prenorm(xx, ln, lin, cache, dtype, activation, residual, col_scale)
    if mode["m"] == "C":
        real_ln(xx, cache.get(ln.weight, dtype), cache.get(ln.bias, dtype), ln.eps)
    return _nn.linear(xx, lin, cache, dtype, activation, residual, col_scale=col_scale)


def ln_passthrough(xx, w, b=None, eps=1e-5, residual=None, residual_alpha=1.0, return_sum=False, out_blocked=False):
    if mode["m"] == "A" or out_blocked or residual is not None:
        return real_ln(xx, w, b, eps, residual, residual_alpha, return_sum, out_blocked)
    if mode["m"] == "C":
        real_ln(xx, w, b, eps)
    return xx


fa_mod.prenorm_linear = prenorm
_nn.prenorm_linear = prenorm
# FusedMLP.forward calls ops.layernorm itself (blocked or plain) and then ops.fused_mlp: patch its `ops.layernorm` view
class _Ops:
    def __getattr__(self, k):
        return getattr(ops, k)
    def layernorm(self, xx, w, b=None, eps=1e-5, residual=None, residual_alpha=1.0, return_sum=False, out_blocked=False):
        if mode["m"] == "A":
            return real_ln(xx, w, b, eps, residual, residual_alpha, return_sum, out_blocked)
        if mode["m"] == "C":
            real_ln(xx, w, b, eps)
        return xx
    def fused_mlp(self, xx, *a, x_blocked_shape=None, **kw):
        return ops.fused_mlp(xx, *a, x_blocked_shape=(x_blocked_shape if mode["m"] == "A" else None), **kw)
mlp_mod.ops = _Ops()


def run(m, steps=10):
    mode["m"] = m
    with torch.no_grad():
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


res = {"A": [], "B": [], "C": []}
for rd in range(3):
    for m in ("A", "B", "C"):
        res[m].append(run(m))
for m in ("A", "B", "C"):
    print(m, "ms/step", " ".join(f"{t:.3f}" for t in res[m]), "median", f"{sorted(res[m])[1]:.3f}")
a, b, c = (sorted(res[m])[1] for m in ("A", "B", "C"))
print(f"fold best case B - A = {b - a:+.3f} ms ({(b - a) / a * 100:+.2f} %), with a statistics pass C - A = {c - a:+.3f} ms ({(c - a) / a * 100:+.2f} %)")
