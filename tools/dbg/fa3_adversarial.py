import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
torch.manual_seed(1)
B, S, H, D = 1, 1024, 8, 64
for dtype in (torch.float16, torch.bfloat16):
    for name, qs, ks in (("x20", 20.0, 1.0), ("tiny", 1e-3, 1e-3), ("const", 0.0, 1.0), ("ramp", None, None)):
        q = torch.randn(B, S, H, D, device="cuda") * (qs if qs is not None else 1.0)
        k = torch.randn(B, S, H, D, device="cuda") * (ks if ks is not None else 1.0)
        if name == "ramp":  # scores increase with the key index: the reference moves in (almost) every tile
            k = k.abs() * torch.linspace(0.1, 6.0, S, device="cuda")[None, :, None, None]
            q = q.abs()
        q, k = q.to(dtype), k.to(dtype)
        v = torch.randn(B, S, H, D, device="cuda").to(dtype)
        for causal in (False, True):
            o, lse = ops.fa3_fwd(q, k, v, causal=causal, return_lse=True)
            qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
            s = qf @ kf.transpose(-1, -2) / 8.0
            if causal:
                s = s.masked_fill(torch.triu(torch.ones(S, S, device="cuda", dtype=torch.bool), 1), float("-inf"))
            ref = (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3)
            rel = ((o.float() - ref).abs().mean() / ref.abs().mean()).item()
            print(str(dtype)[6:], name, "causal" if causal else "full", "rel", f"{rel:.2e}", "nan", int(torch.isnan(o.float()).sum()),
                  "lse err", f"{(lse - torch.logsumexp(s, -1)).abs().max().item():.2e}", "max score", f"{s[~torch.isinf(s)].max().item():.1f}")
