#!/usr/bin/env python3
"""A/B timing of GEMM shapes (C2 layer) for the current MIO_GEMM_VAR / MIO_GEMM_IMPL environment."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
from tools.kbench import timeit
M, d, I, dt, dev = 32768, 1024, 4096, torch.bfloat16, "cuda"
torch.manual_seed(0)
SC = float(os.environ.get("MIO_AB_SCALE", "1"))  # 0 -> all-zero operands (low-power reference point)
res = []
for name, N, K in (("qkv", 3 * d, d), ("oproj", d, d), ("fc2", d, I)):
    x = torch.randn(M, K, device=dev, dtype=dt) * SC
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    t = timeit(lambda: ops.gemm_bias_act(x, w, None, out=out), 20)
    res.append(f"{name} {t*1e3:.3f}ms {2*M*N*K/t/1e12:.0f}TF")
x = torch.randn(M, d, device=dev, dtype=dt) * SC
w = (torch.randn(I, d, device=dev) * 0.02).to(dt)
b = (torch.randn(I, device=dev) * 0.02).to(dt)
out = torch.empty(M, I, device=dev, dtype=dt)
t = timeit(lambda: ops.gemm_bias_act(x, w, b, "gelu", out=out), 20)
res.append(f"fc1+gelu {t*1e3:.3f}ms {2*M*I*d/t/1e12:.0f}TF")
print(os.environ.get("MIO_GEMM_VAR", "0"), os.environ.get("MIO_GEMM_IMPL", "default"), " | ".join(res))
