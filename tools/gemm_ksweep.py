#!/usr/bin/env python3
"""Fixed cost vs per-K-tile cost of the GEMM kernels: M=N=4096 (one 256x256 tile per CU), K swept."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
from tools.kbench import timeit
dt, dev = torch.bfloat16, "cuda"
torch.manual_seed(0)
M = N = 4096
res = []
for K in (32, 64, 512, 1024, 2048, 4096, 8192):
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm_bias_act(x, w, None, out=out), 20)
    tt = timeit(lambda: torch.nn.functional.linear(x, w), 20)
    res.append(f"K{K}: {t*1e6:.1f}us (blaslt {tt*1e6:.1f}us)")
print(os.environ.get("MIO_GEMM_IMPL", "4w"), os.environ.get("MIO_GEMM_VAR", "0"), " | ".join(res))
