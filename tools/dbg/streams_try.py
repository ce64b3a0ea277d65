import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ml-inference-optimizer_amd"))
from mio.synthetic import GPT2ShapedStack
d, H, L, I, B, S = 1024, 16, 24, 4096, 8, 4096
model = GPT2ShapedStack(d, H, L, I, causal=True, precision="bf16", seed=0).to(device="cuda", dtype=torch.bfloat16).eval()
x = torch.randn(B, S, d, device="cuda", dtype=torch.bfloat16)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
xa, xb = x[:4].contiguous(), x[4:].contiguous()
def two():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): ya = model(xa)
    with torch.cuda.stream(s2): yb = model(xb)
    cur.wait_stream(s1); cur.wait_stream(s2)
    return ya, yb
with torch.no_grad():
    print("one stream B=8      ms/step", timeit(lambda: model(x)))
    print("two streams 2 x B=4 ms/step", timeit(two))
    print("one stream 2 x B=4  ms/step", timeit(lambda: (model(xa), model(xb))))
