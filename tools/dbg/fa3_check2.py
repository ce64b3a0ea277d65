import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
torch.manual_seed(0)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for name in ("q0_vrand", "qrand_v1", "qsmall", "full"):
    q, k, v = (torch.randn(1, S, 1, 64, device="cuda", dtype=torch.float16) for _ in range(3))
    if name == "q0_vrand": q.zero_()
    if name == "qrand_v1": v.fill_(1.0)
    if name == "qsmall": q.mul_(0.01)
    o, lse = ops.fa3_fwd(q, k, v, causal=False, return_lse=True)
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    s = qf @ kf.transpose(-1, -2) / 8.0
    ref = (torch.softmax(s, -1) @ vf).permute(0, 2, 1, 3)
    err = (o.float() - ref).abs().nan_to_num(nan=9)
    print(name, "nan", int(torch.isnan(o.float()).sum()), "maxerr", float(err.max()), "lse err", float((lse - torch.logsumexp(s, -1)).abs().nan_to_num(nan=9, posinf=8).max()),
          "o[0,:4,0,:3]", o[0, :4, 0, :3].float().cpu().tolist(), "lse[:4]", lse[0, 0, :4].cpu().tolist(), "ref lse", torch.logsumexp(s, -1)[0, 0, :4].cpu().tolist())
