#!/usr/bin/env python3
"""Kernel micro-benchmarks at the BASELINE config-2 shapes (B8 S4096 d1024 H16, bf16), random data.
Prints achieved TFLOP/s per kernel; used while tuning (not the driver's bench)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops  # noqa: E402


def timeit(fn, iters=20, warmup=3, sustain_s=0.3):
    """Seconds per call, HIP events.  The chip only settles at its sustained clock / power point after a few
    hundred ms of back-to-back work (tools/gemm_power.py: a 20-launch burst measured 217 us for a GEMM that runs
    191 us sustained), so warm up for `sustain_s` of queued launches and time at least as long."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    est = max(s.elapsed_time(e) * 1e-3, 1e-6)
    n = int(sustain_s / est)
    for _ in range(n):
        fn()
    iters = max(iters, n)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--what", default="all")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev, dt = "cuda", torch.bfloat16
    B, S, H, D, d, I = 8, 4096, 16, 64, 1024, 4096
    M = B * S
    torch.manual_seed(0)
    if a.what in ("all", "attn"):
        q, k, v = (torch.randn(B, S, H, D, device=dev, dtype=dt) for _ in range(3))
        for causal in (True, False):
            t = timeit(lambda: ops.fa3_fwd(q, k, v, causal=causal), a.iters)
            fl = (2 if causal else 4) * B * S * S * d * (1 if not causal else 1)
            if causal:
                fl = 2 * B * S * (S + 1) * d
            print(f"fa3_fwd causal={causal}: {t*1e3:.3f} ms  {fl/t/1e12:.1f} TFLOP/s")
        q5, k5, v5 = (torch.randn(B, S, 16, 80, device=dev, dtype=dt) for _ in range(3))
        t = timeit(lambda: ops.fa3_fwd(q5, k5, v5), a.iters)
        print(f"fa3_fwd D=80 noncausal: {t*1e3:.3f} ms  {4*B*S*S*1280/t/1e12:.1f} TFLOP/s")
        q8, k8, v8 = (torch.randn(B, S, 8, 128, device=dev, dtype=dt) for _ in range(3))
        t = timeit(lambda: ops.fa3_fwd(q8, k8, v8, causal=True), a.iters)
        print(f"fa3_fwd D=128 causal: {t*1e3:.3f} ms  {2*B*S*(S+1)*1024/t/1e12:.1f} TFLOP/s")
    if a.what in ("all", "gemm"):
        x = torch.randn(M, d, device=dev, dtype=dt)
        for name, N, K, act in [("qkv", 3 * d, d, "none"), ("oproj", d, d, "none"), ("fc1+gelu", I, d, "gelu"),
                                ("fc2", d, I, "none")]:
            xin = torch.randn(M, K, device=dev, dtype=dt)
            w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
            b = torch.zeros(N, device=dev, dtype=dt)
            out = torch.empty(M, N, device=dev, dtype=dt)
            t = timeit(lambda: ops.gemm_bias_act(xin, w, b, act, out=out), a.iters)
            print(f"gemm {name} M={M} N={N} K={K}: {t*1e3:.3f} ms  {2*M*N*K/t/1e12:.1f} TFLOP/s")
            tt = timeit(lambda: torch.nn.functional.linear(xin, w, b), a.iters)
            print(f"   torch(hipBLASLt) linear: {tt*1e3:.3f} ms  {2*M*N*K/tt/1e12:.1f} TFLOP/s")
        w1 = (torch.randn(I, d, device=dev) * 0.02).to(dt)
        w2 = (torch.randn(d, I, device=dev) * 0.02).to(dt)
        b1 = torch.zeros(I, device=dev, dtype=dt)
        b2 = torch.zeros(d, device=dev, dtype=dt)
        xs = x.view(B, S, d)
        t = timeit(lambda: ops.fused_mlp(xs, w1, b1, w2, b2, "gelu"), a.iters)
        print(f"fused_mlp gelu: {t*1e3:.3f} ms  {4*M*d*I/t/1e12:.1f} TFLOP/s")
        wg = (torch.randn(I, d, device=dev) * 0.02).to(dt)
        t = timeit(lambda: ops.fused_mlp(xs, w1, b1, w2, b2, "swiglu", wg, b1), a.iters)
        print(f"fused_mlp swiglu: {t*1e3:.3f} ms  {6*M*d*I/t/1e12:.1f} TFLOP/s")
    if a.what in ("all", "rows"):
        x = torch.randn(M, d, device=dev, dtype=dt)
        w = torch.ones(d, device=dev, dtype=dt)
        t = timeit(lambda: ops.layernorm(x, w, w), a.iters)
        print(f"layernorm: {t*1e6:.1f} us  {2*M*d*2/t/1e12:.2f} TB/s")
        t = timeit(lambda: ops.layernorm(x, w, w, residual=x, return_sum=True), a.iters)
        print(f"residual+layernorm: {t*1e6:.1f} us  {4*M*d*2/t/1e12:.2f} TB/s")
    if a.what in ("all", "decode"):
        # B 8: K + V = 128 MiB, resident in the 256 MiB Infinity Cache across timed repeats; B 64: 1 GiB, streams from HBM
        for Bd, Hd, Dd in ((8, H, D), (64, H, D), (32, 8, 128)):
            bs, L, ctxlen = 16, 1, 4096
            nblk = Bd * ctxlen // bs
            kc = torch.randn(nblk, L, bs, Hd, Dd, device=dev, dtype=dt)
            vc = torch.randn(nblk, L, bs, Hd, Dd, device=dev, dtype=dt)
            bt = torch.randperm(nblk, device=dev).view(Bd, -1).to(torch.int32)
            cl = torch.full((Bd,), ctxlen, device=dev, dtype=torch.int32)
            q = torch.randn(Bd, Hd, 1, Dd, device=dev, dtype=dt)
            o = torch.empty_like(q)
            t = timeit(lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctxlen, 0), a.iters)
            print(f"paged decode B={Bd} H={Hd} D={Dd} ctx=4096: {t*1e6:.1f} us  {2*Bd*ctxlen*Hd*Dd*2/t/1e12:.2f} TB/s")
            del kc, vc


if __name__ == "__main__":
    main()
