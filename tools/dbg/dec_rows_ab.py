#!/usr/bin/env python3
"""Whole-token-row decode kernel vs the per-head kernel (diagnostic build: mio_dbg_set(6, 1) forces the latter) at the cases
round 3 moved to the row kernel: long contexts at B < 16 (cache streams from HBM) and GQA with 8 queries per kv head."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
from tools.kbench import timeit
dt = torch.bfloat16
for Bd, Hd, Hkv, Dd, ctx in ((8, 16, 16, 64, 32768), (4, 16, 16, 64, 65536), (8, 16, 16, 64, 4096), (32, 32, 4, 128, 8192), (64, 32, 4, 128, 4096),
                             (64, 16, 16, 64, 4096)):
    bs = 16
    nblk = Bd * ctx // bs
    kc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    vc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    bt = torch.randperm(nblk, device="cuda").view(Bd, -1).to(torch.int32)
    cl = torch.full((Bd,), ctx, device="cuda", dtype=torch.int32)
    q = torch.randn(Bd, Hd, 1, Dd, device="cuda", dtype=dt)
    o = torch.empty_like(q)
    res, outs = {0: [], 1: []}, {}
    for rd in range(3):
        for mode in (0, 1):
            _lib.lib.mio_dbg_set(6, mode)
            ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0)
            outs[mode] = o.float().clone()
            res[mode].append(timeit(lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0), 20, sustain_s=0.1))
    nb = 2 * Bd * ctx * Hkv * Dd * 2
    d = (outs[0] - outs[1]).abs().max().item()
    print(f"B={Bd} H={Hd} Hkv={Hkv} D={Dd} ctx={ctx} ({nb/2**20:.0f} MiB): shipped {sorted(res[0])[1]*1e6:.1f} us {nb/sorted(res[0])[1]/1e12:.2f} TB/s | "
          f"per-head {sorted(res[1])[1]*1e6:.1f} us {nb/sorted(res[1])[1]/1e12:.2f} TB/s | max|d| {d:.2e}", flush=True)
    del kc, vc
_lib.lib.mio_dbg_set(6, 0)
