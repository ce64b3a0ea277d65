import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
torch.manual_seed(0)
for (B, H, S, causal) in [(1, 1, 256, False), (1, 1, 256, True), (1, 2, 512, False), (1, 2, 512, True), (2, 4, 300, True), (1, 8, 1024, True)]:
    q, k, v = (torch.randn(B, S, H, 64, device="cuda", dtype=torch.float16) for _ in range(3))
    o, lse = ops.fa3_fwd(q, k, v, causal=causal, return_lse=True)
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) / 8.0
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(S, S, device="cuda", dtype=torch.bool), 1), float("-inf"))
    ref = (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3)
    err = (o.float() - ref).abs()
    bad = torch.isnan(o.float()).any(-1) | (err.max(-1).values > 0.02)
    rows = bad[0, :, 0].nonzero().flatten().tolist()
    print(B, H, S, causal, "nan", int(torch.isnan(o.float()).sum()), "maxerr", float(err.nan_to_num(nan=9).max()), "bad rows(b0,h0)", rows[:10], "...", rows[-5:], len(rows),
          "lse err", float((lse - torch.logsumexp(s, -1)).abs().nan_to_num(nan=9).max()))
