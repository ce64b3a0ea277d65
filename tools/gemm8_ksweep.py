#!/usr/bin/env python3
"""Per-K-tile slope and per-tile intercept of the one-tile GEMM kernels (diagnostic library): time at K = 512 .. 4096 for a fixed
(M, N) with blocked weights; slope = cost of 32 K-tiles of 32, intercept = prologue + epilogue + launch tail per tile."""
import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
sys.path.insert(0, ROOT)
from mio import ops, _lib
from tools.kbench import timeit
impls = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [6, 8, 9]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
xblk = len(sys.argv) > 3 and sys.argv[3] == "xblk"  # activation operand in the blocked layout too
M, N, dt, dev = 32768, 3072, torch.bfloat16, "cuda"
torch.manual_seed(0)
lib = _lib.lib
Ks = (512, 1024, 2048, 4096)
data = {}
for K in Ks:
    x = torch.randn(M, K, device=dev, dtype=dt) * scale
    w = (torch.randn(N, K, device=dev) * 0.02 * scale).to(dt)
    data[K] = (ops.block_weight(x) if xblk else x, w, ops.block_weight(w), torch.empty(M, N, device=dev, dtype=dt))
res = {}
for rd in range(2):
    for impl in impls:
        lib.mio_dbg_set(4, impl)
        for K in Ks:
            x, w, wb, out = data[K]
            t = timeit(lambda: ops.gemm_bias_act(x, w, None, out=out, w_blocked=wb, x_blocked_shape=(M, K) if xblk else None), 20)
            res.setdefault((impl, K), []).append(t)
tiles_per_cu = (M // 256) * (N // 256) / 256
for impl in impls:
    ts = {K: min(res[(impl, K)]) for K in Ks}
    slope = (ts[4096] - ts[1024]) / (3072 / 32) / tiles_per_cu  # seconds per K-tile of 32 per workgroup
    icpt = ts[1024] / tiles_per_cu - slope * 32
    print(f"impl {impl} scale {scale} xblk {xblk}: " + " ".join(f"K{K} {ts[K]*1e3:.4f}ms/{2*M*N*K/ts[K]/1e12:.0f}TF" for K in Ks) +
          f" | per K-tile {slope*1e9:.0f} ns, per-tile overhead {icpt*1e6:.2f} us", flush=True)
lib.mio_dbg_set(4, 0)
