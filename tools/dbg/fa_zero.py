#!/usr/bin/env python3
"""Is the attention kernel power / clock bound?  Same launches on random vs all-zero vs constant operands."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
def run(q, k, v, n):
    o = torch.empty_like(q)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(300): ops.fa3_fwd(q, k, v, causal=True, out=o)
    s.record()
    for _ in range(n): ops.fa3_fwd(q, k, v, causal=True, out=o)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
rnd = [torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
zero = [torch.zeros_like(t) for t in rnd]
ones = [torch.ones_like(t) for t in rnd]
for impl in (3, 4):
    _lib.lib.mio_dbg_set(1, impl)
    for name, (q, k, v) in (("random", rnd), ("zeros", zero), ("ones", ones), ("random q,k / zero v", (rnd[0], rnd[1], zero[2])),
                            ("zero q,k / random v", (zero[0], zero[1], rnd[2]))):
        print(f"fwd{impl} {name:24s} {run(q, k, v, 300):.4f} ms", flush=True)
