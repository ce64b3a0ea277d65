#!/usr/bin/env python3
"""Timing-only ablations of fa3_fwd4_kernel<bf16, causal> (diagnostic build; WRONG results by design), interleaved in one process."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
o = torch.empty_like(q)
NAMES = {0: "full", 1: "-barrier", 2: "-LDS reads", 4: "-exp/cvt", 8: "-scale/max/update", 12: "-exp -scale/max", 16: "-MFMA",
         32: "-DMA", 35: "-barrier -LDS -DMA", 47: "only MFMA (+waits)"}
variants = [int(a) for a in sys.argv[1:]] or sorted(NAMES)
_lib.lib.mio_dbg_set(1, 4)
def run(n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        ops.fa3_fwd(q, k, v, causal=True, out=o)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
res = {a: [] for a in variants}
for a in variants:
    _lib.lib.mio_dbg_set(0, a); run(200)
for r in range(5):
    for a in variants:
        _lib.lib.mio_dbg_set(0, a); run(100); res[a].append(run(300))
base = min(res[variants[0]])
for a in variants:
    t = sorted(res[a])
    print(f"ABL {a:2d} {NAMES.get(a, ''):24s} min {t[0]:.4f} med {t[len(t)//2]:.4f} ms   ({(t[0] / base - 1) * 100:+.1f} %)", flush=True)
