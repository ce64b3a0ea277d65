#!/usr/bin/env python3
"""Timing-only ablations of fa3_fwd5_kernel (diagnostic library; results are wrong by construction): what do the reference
test, the barrier, the DMA wait and the DMA cost per launch?  Random and all-zero operands (the latter: no power cap)."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
rnd = [torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3)]
rnd[1] = (rnd[1].float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
zero = [torch.zeros_like(t) for t in rnd]
def run(q, k, v, n=200):
    o = torch.empty_like(q)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(100): ops.fa3_fwd(q, k, v, causal=True, out=o, k_prescaled=True)
    s.record()
    for _ in range(n): ops.fa3_fwd(q, k, v, causal=True, out=o, k_prescaled=True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
names = {0: "full kernel", 1: "no reference test", 2: "no barrier", 4: "no DMA wait", 8: "no DMA", 15: "none of the four"}
for rep in range(2):
    for bits, nm in names.items():
        _lib.lib.mio_dbg_set(0, bits)
        print(f"{nm:22s} random {run(*rnd):.4f} ms   zeros {run(*zero):.4f} ms", flush=True)
_lib.lib.mio_dbg_set(0, 0)
for rep in range(2):
    for pr, nm in ((1, "no priorities"), (2, "waves 4-7 at priority 1"), (0, "waves 0-3 at priority 1")):
        _lib.lib.mio_dbg_set(3, pr)
        print(f"{nm:26s} random {run(*rnd):.4f} ms   zeros {run(*zero):.4f} ms", flush=True)
_lib.lib.mio_dbg_set(3, 0)
