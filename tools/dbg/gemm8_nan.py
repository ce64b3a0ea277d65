import os, sys
os.environ["MIO_LIB_DBG"] = "1"
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops, _lib
_lib.lib.mio_dbg_set(4, int(sys.argv[1]) if len(sys.argv) > 1 else 8)
torch.manual_seed(0)
for (M, N, K, act, bias) in [(8192, 8192, 512, "gelu", True), (4101, 4096, 512, "gelu_erf", True), (4101, 4104, 1024, "silu", True), (32768, 3072, 1024, "none", True), (8192, 8192, 128, "relu", False), (8200, 8192, 256, "none", True)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
    b = (torch.randn(N, device="cuda") * 0.02).to(torch.bfloat16) if bias else None
    wb = ops.block_weight(w)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = ops.gemm_bias_act(x, w, b, act, out=out, w_blocked=wb).float()
    ref = x.float() @ w.float().T
    if b is not None:
        ref = ref + b.float()
    if act == "gelu":
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if act == "relu":
        ref = torch.relu(ref)
    if act == "silu":
        ref = torch.nn.functional.silu(ref)
    if act == "gelu_erf":
        ref = torch.nn.functional.gelu(ref)
    err = (y - ref).abs().nan_to_num(nan=1e3)
    tm, tn = (M + 255) // 256, (N + 255) // 256
    bad = []
    for i in range(tm):
        for j in range(tn):
            e = err[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].max().item()
            if e > 0.1:
                bad.append((i, j, round(e, 2)))
    print(M, N, K, act, bias, "tiles", tm * tn, "bad tiles", len(bad), bad[:8], flush=True)
    if bad:
        i, j, _ = bad[0]
        e = err[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        rows = (e > 0.1).any(1).nonzero().flatten().tolist()
        cols = (e > 0.1).any(0).nonzero().flatten().tolist()
        print("  first bad tile rows", rows[:8], "...", rows[-4:], "n", len(rows), "cols", cols[:8], "...", cols[-4:], "n", len(cols))
        print("  nan count", torch.isnan(y).sum().item(), "of", y.numel())
