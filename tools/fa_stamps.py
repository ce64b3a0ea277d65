#!/usr/bin/env python3
"""In-kernel phase stamps of fa3_fwd3_kernel (MIO_FA_DBG_PTR + MIO_FA_IMPL=3): cycles per KV tile spent in the DMA
issue, phase 1 (QK^T || exp), phase 2 (PV || max), the reference update and the wait + barrier."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")  # A/B switches and stamp kernels live in libmio_hip_dbg.so (make dbg)
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
causal = (sys.argv[1] == "causal") if len(sys.argv) > 1 else False
B, S, H, D = 8, 4096, 16, 64
nblk = (S + 255) // 256
grid = B * H * ((nblk + 1) // 2 if causal else nblk)
dbg = torch.zeros(grid * 4 * 16, dtype=torch.int64, device="cuda")
os.environ["MIO_FA_DBG_PTR"] = str(dbg.data_ptr())
os.environ["MIO_FA_IMPL"] = "3"
from mio import ops
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
for _ in range(300):
    ops.fa3_fwd(q, k, v, causal=causal)
torch.cuda.synchronize()
d = dbg.view(grid, 4, 16).cpu().double()
d = d[d[:, 0, 7] > 0]  # persistent launch: only gridDim.x workgroups write a record
print(f"{d.shape[0]} workgroups, {d[..., 6].mean().item():.0f} tiles walked per workgroup")
nw = d[..., 5].clamp_min(1)
names = ["phase1 QK||exp", "edge masks", "phase2 PV||max||dma", "update", "wait+barrier"]
tot = 0
for i, n in enumerate(names):
    per = (d[..., i] / nw).mean().item()
    tot += per
    print(f"{n:16s} {per:8.0f} cycles per tile (per wave, mean over waves; both passes)")
print(f"{'sum':16s} {tot:8.0f}")
life = d[..., 7].mean().item()
inloop = d[..., :5].sum(-1).mean().item()
print(f"workgroup lifetime {life:9.0f} cycles, of which in the tile loop {inloop:9.0f} ({100 * inloop / life:.1f} %); "
      f"outside (Q load, first tiles, epilogue, helper iterations) {life - inloop:9.0f}")
for i, n in enumerate(["Q load + state init", "first K/V tiles land", "tile 0 scores/max", "tile loop + helpers + drain", "epilogue"]):
    print(f"  {n:28s} {d[..., 8 + i].mean().item():9.0f} cycles per workgroup (all passes)")
for w in range(4):
    print(f"  wave {w}: " + "  ".join(f"{(d[:, w, i] / nw[:, w]).mean().item():7.0f}" for i in range(5)), " n_w", nw[:, w].mean().item())
