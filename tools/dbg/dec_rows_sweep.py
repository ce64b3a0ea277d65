#!/usr/bin/env python3
"""Workgroup-count sweep of the whole-token-row decode kernel (diagnostic build: mio_dbg_set(2, target))."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
from tools.kbench import timeit
dt = torch.bfloat16
for Bd, Hd, Dd in ((8, 16, 64), (64, 16, 64), (32, 8, 128), (256, 16, 64)):
    bs, ctx = 16, 4096
    nblk = Bd * ctx // bs
    kc = torch.randn(nblk, 1, bs, Hd, Dd, device="cuda", dtype=dt)
    vc = torch.randn(nblk, 1, bs, Hd, Dd, device="cuda", dtype=dt)
    bt = torch.randperm(nblk, device="cuda").view(Bd, -1).to(torch.int32)
    cl = torch.full((Bd,), ctx, device="cuda", dtype=torch.int32)
    q = torch.randn(Bd, Hd, 1, Dd, device="cuda", dtype=dt)
    o = torch.empty_like(q)
    for target in (256, 512, 1024, 2048, 4096):
        _lib.lib.mio_dbg_set(2, target)
        t = timeit(lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0), 30)
        print(f"B={Bd} H={Hd} D={Dd} target {target}: {t*1e6:.1f} us  {2*Bd*ctx*Hd*Dd*2/t/1e12:.2f} TB/s", flush=True)
    del kc, vc
