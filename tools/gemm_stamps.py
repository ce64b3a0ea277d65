#!/usr/bin/env python3
"""In-kernel stamps of the 4-wave GEMM (diagnostic build path, MIO_GEMM_DBG_PTR): per-wave prologue / loop /
epilogue cycles and the in-kernel clock (s_memtime / s_memrealtime)."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
M = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
NT = ((M + 255) // 256) * ((N + 255) // 256)
dt, dev = torch.bfloat16, "cuda"
dbg = torch.zeros(NT * 4 * 8, dtype=torch.int64, device=dev)
os.environ["MIO_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from mio import ops
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=dt)
w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(5):
    ops.gemm_bias_act(x, w, None, out=out)
torch.cuda.synchronize()
d = dbg.view(NT, 4, 8).cpu().double()
pro = d[..., 1] - d[..., 0]
loop = d[..., 2] - d[..., 1]
epi = d[..., 3] - d[..., 2]
tot = d[..., 3] - d[..., 0]
real = (d[..., 5] - d[..., 4]) / 100.0  # us (100 MHz)
clk = tot / real / 1e3  # GHz
print(f"M={M} N={N} K={K}: prologue {pro.mean():.0f} cyc, loop {loop.mean():.0f} (min {loop.min():.0f} max {loop.max():.0f}), "
      f"epilogue {epi.mean():.0f}, total {tot.mean():.0f} cyc = {real.mean():.1f} us, clock {clk.mean():.2f} GHz")
nmf = (K // 32) * 64
print(f"   cycles per 16x16x32 MFMA in loop: {loop.mean() / nmf:.1f}   start spread {(d[..., 4].max() - d[..., 4].min()) / 100:.1f} us "
      f"end spread {(d[..., 5].max() - d[..., 5].min()) / 100:.1f} us  kernel span {(d[..., 5].max() - d[..., 4].min()) / 100:.1f} us")
