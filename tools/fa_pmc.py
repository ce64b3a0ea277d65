#!/usr/bin/env python3
"""Back-to-back launches of the attention kernel at the benchmark shape, for rocprofv3 --pmc / --kernel-trace runs.
usage: fa_pmc.py [3|4] [n]   (4 = fa3_fwd4_kernel through the diagnostic library)"""
import os, sys
impl = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if impl == 4:
    os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
if impl == 4:
    _lib.lib.mio_dbg_set(1, 4)
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
o = torch.empty_like(q)
for _ in range(n):
    ops.fa3_fwd(q, k, v, causal=True, out=o)
torch.cuda.synchronize()
