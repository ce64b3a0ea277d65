#!/usr/bin/env python3
"""Same-process A/B of the GEMM pipelines on the C2 layer shapes (diagnostic library: mio_dbg_set(4, impl)).
impl 0 = shipped dispatch, 6 = gemm4w16 one-tile, 8 / 9 = gemm8w (two / one phase per K-tile), ...
Checks every variant against an fp32 matmul first, then times them interleaved (rounds x variants)."""
import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops, _lib
sys.path.insert(0, ROOT)
from tools.kbench import timeit

impls = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 8, 9]
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
xblk = len(sys.argv) > 3 and sys.argv[3] == "xblk"  # activation operand in the blocked layout (as on the main path)
M, d, I, dt, dev = 32768, 1024, 4096, torch.bfloat16, "cuda"
torch.manual_seed(0)
lib = _lib.lib
shapes = [("qkv", 3 * d, d, "none", False), ("oproj", d, d, "none", True), ("fc1+gelu", I, d, "gelu", False),
          ("fc2", d, I, "none", True)]
data = {}
for name, N, K, act, res in shapes:
    x = torch.randn(M, K, device=dev, dtype=dt)
    w = (torch.randn(N, K, device=dev) * 0.02).to(dt)
    b = (torch.randn(N, device=dev) * 0.02).to(dt)
    r = torch.randn(M, N, device=dev, dtype=dt) if res else None
    wb = ops.block_weight(w)
    out = torch.empty(M, N, device=dev, dtype=dt)
    data[name] = (x, w, b, r, wb, out, ops.block_weight(x) if xblk else None)

def run(name, N, K, act, res):
    x, w, b, r, wb, out, xb = data[name]
    if xb is not None:
        return ops.gemm_bias_act(xb, w, b, act, residual=r, out=out, w_blocked=wb, x_blocked_shape=(M, K))
    return ops.gemm_bias_act(x, w, b, act, residual=r, out=out, w_blocked=wb)

# correctness: sampled rows against fp32 (all columns), every tile row touched
for impl in impls:
    lib.mio_dbg_set(4, impl)
    for name, N, K, act, res in shapes:
        x, w, b, r, wb, out, _ = data[name]
        out.fill_(float("nan"))
        y = run(name, N, K, act, res)
        torch.cuda.synchronize()
        rows = torch.arange(0, M, 37, device=dev)
        ref = x[rows].float() @ w.float().T + b.float()
        if act == "gelu":
            ref = torch.nn.functional.gelu(ref, approximate="tanh")
        if r is not None:
            ref = ref + r[rows].float()
        err = (y[rows].float() - ref).abs().max().item()
        nan = torch.isnan(y).any().item()
        rel = ((y[rows].float() - ref).abs().mean() / ref.abs().mean()).item()
        print(f"check impl {impl} {name}: max|d| {err:.4f} rel {rel:.2e} nan {nan}", flush=True)

res_t = {}
for rd in range(rounds):
    for impl in impls:
        lib.mio_dbg_set(4, impl)
        for name, N, K, act, res in shapes:
            t = timeit(lambda: run(name, N, K, act, res), 20)
            res_t.setdefault((impl, name), []).append(t)
for name, N, K, act, res in shapes:
    for impl in impls:
        ts = sorted(res_t[(impl, name)])
        med = ts[len(ts) // 2]
        print(f"{name:9s} impl {impl}: median {med*1e3:.4f} ms  min {ts[0]*1e3:.4f}  {2*M*N*K/med/1e12:.0f} TF", flush=True)
lib.mio_dbg_set(4, 0)
