#!/usr/bin/env python3
"""Would decode_gqa_kernel (matrix core) also serve 1..4 query vectors per kv head?  Diagnostic build: mio_dbg_set(6, 3)
forces it; compared with the shipped choice (rows / per-head kernels) over MHA and narrow-GQA shapes."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops, _lib
from tools.kbench import timeit
dt = torch.bfloat16
lib = _lib.lib
cases = ((64, 16, 8, 64, 4096, 1), (64, 32, 16, 128, 4096, 1), (8, 16, 8, 64, 4096, 1), (16, 32, 16, 128, 8192, 1), (64, 16, 16, 64, 4096, 2),
         (64, 24, 8, 128, 4096, 1), (1, 32, 32, 128, 65536, 1), (2, 16, 16, 64, 65536, 1), (1, 16, 8, 64, 131072, 1), (64, 8, 8, 128, 4096, 3))
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
    for mode, name in ((0, "shipped"), (3, "matrix-core")):
        lib.mio_dbg_set(6, mode)
        fn()
        outs[mode] = o.float().clone()
        t = sorted(timeit(fn, 20, sustain_s=0.1) for _ in range(3))[1]
        line += f" {name} {t*1e6:.1f} us {nb/t/1e12:.2f} TB/s |"
    lib.mio_dbg_set(6, 0)
    line += f" max|d| {(outs[0]-outs[3]).abs().max().item():.2e}"
    print(line, flush=True)
    del kc, vc
