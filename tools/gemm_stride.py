#!/usr/bin/env python3
"""Does the power-of-two row stride hurt (L2 channel conflicts)?  Same GEMM with padded leading dimensions."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
from tools.kbench import timeit
dt, dev = torch.bfloat16, "cuda"
torch.manual_seed(0)
for (M, N, K) in ((32768, 3072, 1024), (32768, 1024, 4096), (8192, 8192, 8192), (4096, 4096, 4096)):
    for pad in (0, 64, 192):
        xs = torch.randn(M, K + pad, device=dev, dtype=dt)
        ws = (torch.randn(N, K + pad, device=dev) * 0.02).to(dt)
        x, w = xs[:, :K], ws[:, :K]
        out = torch.empty(M, N, device=dev, dtype=dt)
        t = timeit(lambda: ops.gemm_bias_act(x, w, None, out=out), 10)
        msg = f"M{M} N{N} K{K} pad{pad}: {t*1e3:.3f}ms {2*M*N*K/t/1e12:.0f}TF"
        if pad == 0:
            tt = timeit(lambda: torch.nn.functional.linear(x, w), 10)
            msg += f"   [hipBLASLt {2*M*N*K/tt/1e12:.0f}TF]"
        print(msg)
