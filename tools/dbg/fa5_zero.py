#!/usr/bin/env python3
"""fa3_fwd5_kernel (k_prescaled): the same launches on random vs all-zero operands -- how much of the time is the power cap?"""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
def run(q, k, v, n, causal):
    o = torch.empty_like(q)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(300): ops.fa3_fwd(q, k, v, causal=causal, out=o, k_prescaled=True)
    s.record()
    for _ in range(n): ops.fa3_fwd(q, k, v, causal=causal, out=o, k_prescaled=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
rnd = [torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
rnd[1] = (rnd[1].float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
zero = [torch.zeros_like(t) for t in rnd]
for causal in (True, False):
    for impl, nm in ((0, "fwd5"), (4, "fwd4"), (3, "fwd3")):
        _lib.lib.mio_dbg_set(1, impl)
        for name, (q, k, v) in (("random", rnd), ("zeros", zero), ("random q,k / zero v", (rnd[0], rnd[1], zero[2])),
                                ("zero q,k / random v", (zero[0], zero[1], rnd[2]))):
            print(f"causal={int(causal)} {nm} k_prescaled {name:24s} {run(q, k, v, 300, causal):.4f} ms", flush=True)
