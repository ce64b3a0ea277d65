#!/usr/bin/env python3
"""decode_gqa_kernel (matrix core, shipped for 5..16 query vectors per kv head) against the vector-ALU kernels it replaces
(diagnostic build: mio_dbg_set(6, 2) = whole-token-row kernel where it applies, 6, 1 = per-head kernel), and a sweep of the
workgroup target (mio_dbg_set(2, n))."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
from tools.kbench import timeit
dt = torch.bfloat16
lib = _lib.lib
cases = ((64, 32, 4, 128, 4096, 1), (32, 32, 4, 128, 8192, 1), (8, 32, 4, 128, 32768, 1), (64, 64, 8, 128, 4096, 1), (64, 32, 8, 64, 4096, 2),
         (256, 32, 4, 128, 1024, 1), (16, 128, 8, 128, 4096, 1))
for Bd, Hd, Hkv, Dd, ctx, ql in cases:
    bs = 16
    nblk = Bd * ctx // bs
    kc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    vc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    bt = torch.randperm(nblk, device="cuda").view(Bd, -1).to(torch.int32)
    cl = torch.full((Bd,), ctx, device="cuda", dtype=torch.int32)
    q = torch.randn(Bd, Hd, ql, Dd, device="cuda", dtype=dt)
    o = torch.empty_like(q)
    fn = lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0)
    nb = 2 * Bd * ctx * Hkv * Dd * 2
    line = f"B={Bd} H={Hd} Hkv={Hkv} D={Dd} ctx={ctx} q_len={ql} ({nb/2**20:.0f} MiB):"
    outs = {}
    for mode, name in ((0, "gqa"), (2, "rows/per-head"), (1, "per-head")):
        lib.mio_dbg_set(6, mode)
        fn()
        outs[mode] = o.float().clone()
        t = sorted(timeit(fn, 20, sustain_s=0.1) for _ in range(3))[1]
        line += f" {name} {t*1e6:.1f} us {nb/t/1e12:.2f} TB/s |"
    lib.mio_dbg_set(6, 0)
    line += f" max|gqa - per-head| {(outs[0]-outs[1]).abs().max().item():.2e}"
    print(line, flush=True)
    sw = []
    for target in (256, 512, 768, 1024, 2048):
        lib.mio_dbg_set(2, target)
        fn()
        t = sorted(timeit(fn, 20, sustain_s=0.05) for _ in range(3))[1]
        sw.append(f"{target}:{nb/t/1e12:.2f}")
    lib.mio_dbg_set(2, 0)
    print("   workgroup target sweep (TB/s):", " ".join(sw), flush=True)
    del kc, vc
