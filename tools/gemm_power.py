#!/usr/bin/env python3
"""Clock / power the chip holds while one GEMM shape runs back to back (rocm-smi polled from the host while ~4 s
of launches are queued).  Tells a power-limited kernel (clock sinks as the schedule gets denser, time does not
move) from an issue-limited one.   MIO_AB_SCALE=0 -> zero operands;  MIO_PW_LIB=1 -> torch.matmul (hipBLASLt)."""
import os, subprocess, sys, time
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops

M, d, dt, dev = 32768, 1024, torch.bfloat16, "cuda"
N, K = (int(os.environ.get("MIO_PW_N", 3 * d)), int(os.environ.get("MIO_PW_K", d)))
SC = float(os.environ.get("MIO_AB_SCALE", "1"))
LIB = os.environ.get("MIO_PW_LIB", "0") == "1"
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=dt) * SC
w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
out = torch.empty(M, N, device=dev, dtype=dt)
wt = w.t()
fn = (lambda: torch.matmul(x, wt, out=out)) if LIB else (lambda: ops.gemm_bias_act(x, w, None, out=out))
for _ in range(20):
    fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 16000
s.record()
for _ in range(n):
    fn()
e.record()
samples = []
t0 = time.time()
while not e.query() and time.time() - t0 < 30:
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = [l.split("(")[-1].rstrip(")") for l in r.splitlines() if "sclk" in l]
    pw = [l.split(":")[-1].strip() for l in r.splitlines() if "Power" in l and "W)" in l]
    samples.append((sclk[0] if sclk else "?", pw[0] if pw else "?"))
torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
print(("hipBLASLt" if LIB else os.environ.get("MIO_GEMM_IMPL", "default")), f"N={N} K={K} scale={SC}",
      f"{ms*1e3:.1f} us {2*M*N*K/ms/1e9:.0f} TF", "samples(sclk,W):", samples[1:-1][:6])
