#!/usr/bin/env python3
"""Back-to-back launches of ONE attention structure at the benchmark shape, for rocprofv3 --pmc / --kernel-trace runs
(always through the diagnostic library, which carries every structure).
usage: fa_pmc.py <mode> [n]
  5   fa3_fwd5_kernel, pre-scaled K (what bench.py's stack runs)      5p  fa3_fwd5_kernel, plain K (ops.flash_attention)
  4   fa3_fwd4_kernel (mio_dbg_set(1, 4))                              3   fa3_fwd3_kernel (mio_dbg_set(1, 3))"""
import os, sys
os.environ["MIO_LIB_DBG"] = "1"
mode = sys.argv[1] if len(sys.argv) > 1 else "5"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
kpre = mode == "5"
if kpre:  # K as the QKV projection's column-scale epilogue hands it over
    k = (k.float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
if mode in ("3", "4"):
    _lib.lib.mio_dbg_set(1, int(mode))
o = torch.empty_like(q)
for _ in range(n):
    ops.fa3_fwd(q, k, v, causal=True, out=o, k_prescaled=kpre)
torch.cuda.synchronize()
