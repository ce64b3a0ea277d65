#!/usr/bin/env python3
"""In-kernel phase stamps of fa3_fwd5_kernel (diagnostic library, MIO_FA_DBG_PTR): cycles per KV tile and wave spent in
phase 1 (QK^T || exp; its head = K reads until the first two MFMAs have delivered), the reference test / masks / DMA issue,
phase 2 (PV) and the wait + barrier."""
import os, sys
os.environ.setdefault("MIO_LIB_DBG", "1")
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
causal = (sys.argv[1] == "causal") if len(sys.argv) > 1 else True
B, S, H, D = 8, 4096, 16, 64
nblk = (S + 255) // 256
grid = B * H * ((nblk + 1) // 2 if causal else nblk)
dbg = torch.zeros(grid * 8 * 16, dtype=torch.int64, device="cuda")
os.environ["MIO_FA_DBG_PTR"] = str(dbg.data_ptr())
from mio import ops
torch.manual_seed(0)
q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
k = (k.float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
for _ in range(200):
    ops.fa3_fwd(q, k, v, causal=causal, k_prescaled=True)
torch.cuda.synchronize()
d = dbg.view(grid, 8, 16).cpu().double()
print(f"{d.shape[0]} workgroups, {d[..., 6].mean().item():.0f} tiles walked per workgroup")
nw = d[..., 5].clamp_min(1)
names = ["phase1 QK||exp", "test/mask/dma", "phase2 PV", "wait+barrier", "(phase1 head)"]
tot = 0
for i, n in enumerate(names):
    per = (d[..., i] / nw).mean().item()
    if i < 4:
        tot += per
    print(f"{n:16s} {per:8.0f} cycles per tile (per wave, mean over waves; both passes)")
print(f"{'sum':16s} {tot:8.0f}")
life = d[..., 7].mean().item()
inloop = d[..., :4].sum(-1).mean().item()
print(f"wave lifetime {life:9.0f} cycles, of which in the tile loop {inloop:9.0f} ({100 * inloop / life:.1f} %)")
for i, n in enumerate(["Q load + first tiles land", "tile 0 scores/ref", "tile loop + helpers + drain", "epilogue"]):
    print(f"  {n:28s} {d[..., 8 + i].mean().item():9.0f} cycles per wave (all passes)")
for w in range(8):
    print(f"  wave {w}: " + "  ".join(f"{(d[:, w, i] / nw[:, w]).mean().item():7.0f}" for i in range(5)), " n_w", nw[:, w].mean().item(),
          " life", d[:, w, 7].mean().item())
