#!/usr/bin/env python3
"""In-kernel stamps of the persistent GEMM (MIO_GEMM_DBG_PTR): per (tile, wave) prologue / K loop / read-out
cycles after `sustain` launches, i.e. at the clock the chip holds under sustained load."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
sustain = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
NT = ((M + 255) // 256) * ((N + 255) // 256)
dt, dev = torch.bfloat16, "cuda"
dbg = torch.zeros(NT * 4 * 8, dtype=torch.int64, device=dev)
os.environ["MIO_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from mio import ops
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=dt)
w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(sustain):
    ops.gemm_bias_act(x, w, None, out=out)
torch.cuda.synchronize()
d = dbg.view(NT, 4, 8).cpu().double()
pro, loop, rd = d[..., 1] - d[..., 0], d[..., 2] - d[..., 1], d[..., 3] - d[..., 2]
tot = d[..., 3] - d[..., 0]
real = (d[..., 5] - d[..., 4]) / 100.0
print(f"M={M} N={N} K={K}: per tile: prologue {pro.mean():.0f}, loop {loop.mean():.0f} (min {loop.min():.0f} max {loop.max():.0f}), "
      f"read-out {rd.mean():.0f}, total {tot.mean():.0f} cyc = {real.mean():.2f} us, clock {(tot / real / 1e3).mean():.2f} GHz")
print(f"   cycles per MFMA in loop {loop.mean() / ((K // 32) * 64):.2f};  kernel span {(d[..., 5].max() - d[..., 4].min()) / 100:.1f} us")
wg = d[..., 0, 6]
for b in (0, 1, 100, 255):
    t = (wg == b).nonzero().flatten().tolist()
    seq = sorted(t, key=lambda i: d[i, 0, 0].item())
    print(f"   wg {b}: tiles {seq}: starts (us rel) {[round((d[i,0,4].item()-d[:, :, 4].min().item())/100,1) for i in seq]}"
          f" loop {[int(loop[i,0].item()) for i in seq]} pro {[int(pro[i,0].item()) for i in seq]} rd {[int(rd[i,0].item()) for i in seq]}")
