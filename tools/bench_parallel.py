"""Multi-GPU exchange-path measurements used by bench.py's "extra" block (configs 3 and 4 of BASELINE.json):
tensor-parallel GPT-2-shaped stack (RCCL all-reduce overlapped with the row-parallel GEMM) and ring
attention at S=65536 (K/V exchange over xGMI overlapped with the attention kernel)."""
import time

import torch
import torch.distributed as dist
import torch.nn as nn


def _sync():
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(t):
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    x = torch.tensor([t], device=dev, dtype=torch.float64)
    dist.all_reduce(x, op=dist.ReduceOp.MAX)
    return float(x.item())


class _TPBlock(nn.Module):
    def __init__(self, d, H, I, cfg):
        super().__init__()
        from mio.parallelism import TensorParallelAttention, TensorParallelMLP
        from mio.synthetic import FusedLayerNorm

        self.ln_1, self.ln_2 = FusedLayerNorm(d), FusedLayerNorm(d)
        self.attn = TensorParallelAttention(d, H, cfg, causal=True)
        self.mlp = TensorParallelMLP(d, I, cfg, activation="gelu")

    def forward(self, x):
        x = self.attn(self.ln_1(x), residual=x)
        return self.mlp(self.ln_2(x), residual=x)


def bench_tp(N, B, S, d, H, I, L, dt, steps=3, warmup=1):
    """Same global work as the single-GPU benchmark (B sequences), weights sharded tp=N."""
    from mio.parallelism import TensorParallelConfig

    cfg = TensorParallelConfig(world_size=N, tp_size=N, overlap_chunks=4)
    torch.manual_seed(0)
    blocks = nn.ModuleList([_TPBlock(d, H, I, cfg) for _ in range(L)]).to(device="cuda", dtype=dt).eval()
    with torch.no_grad():
        for m in blocks.modules():
            if isinstance(m, nn.LayerNorm):
                continue
            w, b = getattr(m, "weight", None), getattr(m, "bias", None)
            if isinstance(w, torch.Tensor) and w.dim() == 2:
                w.normal_(0.0, 0.02)
                if isinstance(b, torch.Tensor):
                    b.zero_()
    x = torch.randn(B, S, d, device="cuda", dtype=dt)
    dist.broadcast(x, 0)

    def fwd():
        y = x
        for b in blocks:
            y = b(y)
        return y

    out = {}
    for chunks in (4, 1):
        cfg.overlap_chunks = chunks
        with torch.no_grad():
            for _ in range(warmup):
                fwd()
            _sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                fwd()
            _sync()
            el = _max_over_ranks(time.perf_counter() - t0)
        key = "overlapped" if chunks > 1 else "unoverlapped"
        out[key] = {"ms_per_step": el / steps * 1e3, "tokens_per_s": B * S * steps / el, "allreduce_chunks": chunks}
    out["config"] = f"tp={N} global_batch={B} seq={S} d={d} h={H} L={L} (strong scaling of the 1-GPU workload)"
    del blocks
    torch.cuda.empty_cache()
    return out


def bench_ring(N, S_total, d, H, dt, steps=3, warmup=1):
    """Ring attention core at S_total tokens, B=1: q/k/v shards [1,H,S/N,D] per rank."""
    from mio.parallelism.sequence_parallel import ring_attention

    D = d // H
    Sl = S_total // N
    torch.manual_seed(100 + dist.get_rank())
    q, k, v = (torch.randn(1, H, Sl, D, device="cuda", dtype=dt) for _ in range(3))
    out = {}
    for name, kw in (("noncausal_mesh", dict(exchange="mesh")), ("noncausal_ring", dict(exchange="ring")),
                     ("causal_zigzag_mesh", dict(exchange="mesh", causal=True, zigzag=True))):
        for _ in range(warmup):
            ring_attention(q, k, v, None, layout="bhsd", **kw)
        _sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ring_attention(q, k, v, None, layout="bhsd", **kw)
        _sync()
        el = _max_over_ranks(time.perf_counter() - t0) / steps
        flops = 4.0 * S_total * S_total * d * (0.5 if "causal" in kw else 1.0)
        out[name] = {"ms": el * 1e3, "tokens_per_s": S_total / el, "tflops_total": flops / el / 1e12}
    out["config"] = f"sp={N} seq={S_total} d={d} h={H} B=1 attention core (q/k/v resident per rank)"
    return out
