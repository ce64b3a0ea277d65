#!/usr/bin/env python3
"""What does the GELU read-out of fc1 cost?  Same GEMM (blocked weights, bias) with act none / gelu / relu, random and zero data."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops
M, d, I, dt, dev = 32768, 1024, 4096, torch.bfloat16, "cuda"
torch.manual_seed(0)
def timeit(f, n=300, warm=300):
    for _ in range(warm): f()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for sc in (1.0, 0.0):
    x = torch.randn(M, d, device=dev, dtype=dt) * sc
    w = (torch.randn(I, d, device=dev) * 0.02 * sc).to(dt)
    b = (torch.randn(I, device=dev) * 0.02).to(dt)
    out = torch.empty(M, I, device=dev, dtype=dt)
    for rep in range(2):
        for act in ("none", "gelu", "relu"):
            t = timeit(lambda: ops.gemm_bias_act(x, w, b, act, out=out))
            print(f"scale {sc} fc1 act={act:5s} {t:.4f} ms  {2*M*I*d/t/1e9:.0f} TFLOP/s", flush=True)
