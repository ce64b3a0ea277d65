#!/usr/bin/env python3
"""One ring step's attention launch ((o_acc, lse) carry, B 1, H 16, 8192 local queries x 8192 keys, D 64: the shard shape of
`bench.py --gpus 8`'s ring leg at S 65536): plain K (fa3_fwd3_kernel) vs pre-scaled K (fa3_fwd5_kernel CARRY)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")]
from mio import ops
B, H, S, D = 1, 16, 8192, 64
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
kt = (k.float() * (D ** -0.5 * 1.4426950408889634)).to(torch.bfloat16)
o_acc = torch.zeros(B, S, H, D, dtype=torch.float32, device="cuda")
lse = torch.full((B, H, S), float("-inf"), device="cuda")
def run(kk, kpre, causal, n=100):
    def f(carry):
        ops.fa3_fwd(q, kk, v, layout="bhsd", causal=causal, q_offset=0, k_offset=0, o_acc=o_acc, lse=lse, carry_in=carry,
                    write_out=False, k_prescaled=kpre)
    f(False)
    for _ in range(30): f(True)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f(True)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for causal in (False, True):
    fl = 4.0 * B * H * S * S * D * (0.5 if causal else 1.0)
    for rep in range(2):
        a = run(k, False, causal); b_ = run(kt, True, causal)
        print(f"causal={int(causal)}: plain K {a:.4f} ms ({fl / a / 1e9:.0f} TFLOP/s)   pre-scaled K {b_:.4f} ms ({fl / b_ / 1e9:.0f} TFLOP/s)", flush=True)
