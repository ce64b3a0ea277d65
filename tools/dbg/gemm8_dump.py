import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops, _lib
_lib.lib.mio_dbg_set(4, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
M, N, K = 8192, 8192, 256
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
wb = ops.block_weight(w)
out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
y = ops.gemm_bias_act(x, w, None, "none", out=out, w_blocked=wb)
ref = (x.float() @ w.float().T)
refb = ref.to(torch.bfloat16)
bad = (y.float() - ref).abs() > 0.1
bad |= torch.isnan(y)
print("bad count", bad.sum().item())
idx = bad.nonzero()[:12]
yi = y.view(torch.int16)
ri = refb.view(torch.int16)
for r, c in idx.tolist():
    got = y[r, c].item(); exp = ref[r, c].item()
    # where else in the reference tile does the got bit pattern appear?
    t0r, t0c = r // 256 * 256, c // 256 * 256
    tile = ri[t0r:t0r + 256, t0c:t0c + 256]
    hits = (tile == yi[r, c]).nonzero()[:4].tolist()
    print(f"({r},{c}) got {got:.4e} bits {yi[r,c].item() & 0xffff:04x} exp {exp:.4f} bits {ri[r,c].item() & 0xffff:04x} same-bits-in-ref-tile at {hits}")
# neighbours
r, c = idx[0].tolist()
print("row", r, "cols", c // 8 * 8, "..: got", [f"{v:.3f}" for v in y[r, c // 8 * 8:c // 8 * 8 + 8].float().tolist()])
print("row", r, "cols", c // 8 * 8, "..: exp", [f"{v:.3f}" for v in ref[r, c // 8 * 8:c // 8 * 8 + 8].tolist()])
