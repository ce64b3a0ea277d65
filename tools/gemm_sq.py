#!/usr/bin/env python3
"""One square GEMM (ours and hipBLASLt) for PMC collection."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
dt, dev = torch.bfloat16, "cuda"
torch.manual_seed(0)
x = torch.randn(n, n, device=dev, dtype=dt)
w = (torch.randn(n, n, device=dev) * 0.02).to(dt)
out = torch.empty(n, n, device=dev, dtype=dt)
for _ in range(5):
    ops.gemm_bias_act(x, w, None, out=out)
    torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
