import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio import ops
from tools.kbench import timeit
M, N, K = 32768, 1024, 4096
w = (torch.randn(N, K, device="cuda") * 0.02).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
t = timeit(lambda: ops.gemm_bias_act(x, w, None, out=out))
print(f"x [M,K] row stride 8 KiB (256 MiB): {t*1e3:.3f} ms {2*M*N*K/t/1e12:.0f} TF")
base = torch.randn(M * 64 + K, device="cuda", dtype=torch.bfloat16)
for stride in (64, 512, 1024):
    xs = torch.as_strided(base, (M, K), (stride, 1)) if M * stride + K <= base.numel() else None
    if xs is None:
        base = torch.randn(M * stride + K, device="cuda", dtype=torch.bfloat16); xs = torch.as_strided(base, (M, K), (stride, 1))
    t = timeit(lambda: ops.gemm_bias_act(xs, w, None, out=out))
    print(f"x overlapping rows, row stride {stride*2} B (footprint {M*stride*2/2**20:.0f} MiB): {t*1e3:.3f} ms {2*M*N*K/t/1e12:.0f} TF")
