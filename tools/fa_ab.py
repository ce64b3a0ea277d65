#!/usr/bin/env python3
import os, sys, torch
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
from tools.kbench import timeit
B, S, H, D = 8, 4096, 16, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
t1 = timeit(lambda: ops.fa3_fwd(q, k, v, causal=True), 20)
t2 = timeit(lambda: ops.fa3_fwd(q, k, v, causal=True, q_offset=S), 20)
t3 = timeit(lambda: ops.fa3_fwd(q, k, v, causal=False), 20)
print(os.environ.get("MIO_FA_IMPL", "2"), os.environ.get("MIO_FA_ORDER", "0"), f"causal {t1*1e3:.3f} ms | causal-all-visible {t2*1e3:.3f} | noncausal {t3*1e3:.3f}")
