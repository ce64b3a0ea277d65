import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
torch.manual_seed(0)
for (M, N, K) in [(32768, 3072, 1024), (4096, 4096, 1024), (4133, 4104, 1024), (4133, 4096, 1024), (4096, 4104, 1024), (8192, 2304, 256)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * 0.05).to(torch.bfloat16)
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
    y = ops.gemm_bias_act(x, w, None, out=out).float()
    ref = x.float() @ w.float().T
    err = (y - ref).abs().nan_to_num(nan=1e3)
    tm, tn = (M + 255) // 256, (N + 255) // 256
    bad = []
    for i in range(tm):
        for j in range(tn):
            e = err[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256].max().item()
            if e > 0.1:
                bad.append((i, j, round(e, 2)))
    print(M, N, K, "tiles", tm * tn, "bad tiles", len(bad), bad[:12])
    if bad:
        i, j, _ = bad[0]
        e = err[i * 256:(i + 1) * 256, j * 256:(j + 1) * 256]
        rows = (e > 0.1).any(1).nonzero().flatten().tolist()
        cols = (e > 0.1).any(0).nonzero().flatten().tolist()
        print("  first bad tile rows", rows[:8], "...", rows[-4:], "n", len(rows), "cols", cols[:8], "...", cols[-4:], "n", len(cols))
