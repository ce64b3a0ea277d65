"""Which hipBLASLt kernels torch picks for the benchmark's GEMM shapes (run under rocprofv3 --kernel-trace --stats):
the kernel names carry the macro tile, depth-U, MFMA shape and wave layout."""
import torch
dev = "cuda"
M = 32768
for N, K in ((3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * 0.02
    b = torch.zeros(N, device=dev, dtype=torch.bfloat16)
    for _ in range(400):
        torch.nn.functional.linear(x, w, b)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(200):
        torch.nn.functional.linear(x, w, b)
    ev1.record()
    torch.cuda.synchronize()
    t = ev0.elapsed_time(ev1) / 200
    print(f"N={N} K={K}: {t:.4f} ms  {2*M*N*K/t/1e9:.0f} TFLOP/s", flush=True)
