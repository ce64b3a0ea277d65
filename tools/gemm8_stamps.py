#!/usr/bin/env python3
"""In-kernel stamps of the persistent eight-wave GEMM (diagnostic library, MIO_GEMM_DBG_PTR + mio_dbg_set(4, 8)): where a
tile's time goes at the clock the chip holds under sustained load."""
import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
M = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
N = int(sys.argv[3]) if len(sys.argv) > 3 else 3072
sustain = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
NT = ((M + 255) // 256) * ((N + 255) // 256)
dt, dev = torch.bfloat16, "cuda"
dbg = torch.zeros(NT * 8 * 8, dtype=torch.int64, device=dev)
os.environ["MIO_GEMM_DBG_PTR"] = str(dbg.data_ptr())
from mio import ops, _lib
_lib.lib.mio_dbg_set(4, 8)
torch.manual_seed(0)
x = torch.randn(M, K, device=dev, dtype=dt)
w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
xb, wb = ops.block_weight(x), ops.block_weight(w)
out = torch.empty(M, N, device=dev, dtype=dt)
for _ in range(sustain):
    ops.gemm_bias_act(xb, w, None, out=out, w_blocked=wb, x_blocked_shape=(M, K))
torch.cuda.synchronize()
d = dbg.view(NT, 8, 8).cpu().double()
nk = K // 32
for grp in (0, 1):
    e = d[:, grp * 4:(grp + 1) * 4]
    k0, mid, tail, rd = e[..., 1] - e[..., 0], e[..., 3] - e[..., 1], e[..., 4] - e[..., 3], e[..., 5] - e[..., 4]
    tot = e[..., 5] - e[..., 0]
    real = (e[..., 7] - e[..., 6]) / 100.0
    print(f"group {grp}: K-tile 0 {k0.mean():.0f}, K-tiles 1..{nk-4} {mid.mean():.0f} ({mid.mean()/(nk-4):.0f} each), "
          f"tail (3) {tail.mean():.0f}, read-out {rd.mean():.0f}, total {tot.mean():.0f} cyc = {real.mean():.2f} us, "
          f"clock {(tot / real / 1e3).mean():.2f} GHz")
span = (d[..., 7].max() - d[..., 6].min()) / 100
print(f"M={M} N={N} K={K}: kernel span {span:.1f} us; tiles per workgroup {NT / min(NT, 256):.1f}")
# one workgroup's tiles in time order: start offsets, gaps between read-out end and next tile start
wg0 = [t for t in range(NT) if t % 256 == 0]
t00 = d[:, :, 6].min().item()
for wv in (0, 4):
    print(f"  wg 0 wave {wv}: " + " | ".join(f"start {(d[t, wv, 6].item() - t00) / 100:.1f}us k0 {int((d[t, wv, 1] - d[t, wv, 0]).item())} loop {int((d[t, wv, 4] - d[t, wv, 1]).item())} rd {int((d[t, wv, 5] - d[t, wv, 4]).item())}" for t in wg0))
    gaps = [int((d[wg0[i + 1], wv, 0] - d[wg0[i], wv, 5]).item()) for i in range(len(wg0) - 1)]
    print(f"            cycles between read-out end and the next tile's first stamp: {gaps}")
