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


def bench_tp(N, tp, B, S, d, H, I, L, dt, steps=3, warmup=1):
    """BASELINE config 3: the 1-GPU workload (B sequences of S tokens) per tensor-parallel group of `tp` adjacent ranks
    (N / tp such groups run side by side on different data): column-parallel QKV / fc1, row-parallel out-proj / fc2 +
    all-reduce SUM on the group (reference tensor_parallel.py:299-308), with the all-reduce of row chunk i under the GEMM
    of chunk i + 1 ("overlapped") and as one blocking call after the whole GEMM ("unoverlapped")."""
    from mio.parallelism import TensorParallelConfig

    cfg = TensorParallelConfig(world_size=N, tp_size=tp, overlap_chunks=4)
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
    grp = cfg.get_tp_group()
    dist.broadcast(x, dist.get_global_rank(grp, 0) if grp is not None else 0, group=grp)  # one batch per tp group

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
        out[key] = {"ms_per_step": el / steps * 1e3, "tokens_per_s": (N // tp) * B * S * steps / el,
                    "allreduce_chunks": chunks}
    out["allreduce_payload_MiB_per_layer"] = 2 * B * S * d * 2 / 2 ** 20
    out["config"] = (f"tp={tp} x dp={N // tp}: batch {B} per tp group, seq={S} d={d} h={H} L={L} "
                     f"(strong scaling of the 1-GPU workload inside each group)")
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
    # as SequenceParallelAttention hands it over in ring mode: K pre-scaled by its projection's epilogue where the launches
    # support it (head dim <= 64: fa3_fwd5_kernel with the (o_acc, lse) carry)
    from mio import ops
    kpre = bool(ops.fa3_k_prescaled_ok(1, Sl // 2, Sl // 2, H, D, D, D, carry=True))
    if kpre:
        k = (k.float() * (D ** -0.5 * 1.4426950408889634)).to(dt)
    out = {"k_prescaled": kpre}
    pool = {}
    for name, kw in (("noncausal_mesh", dict(exchange="mesh")),
                     ("noncausal_mesh_unoverlapped", dict(exchange="mesh", overlap=False)),
                     ("noncausal_ring", dict(exchange="ring")),
                     ("noncausal_ring_unoverlapped", dict(exchange="ring", overlap=False)),
                     ("causal_zigzag_mesh", dict(exchange="mesh", causal=True, zigzag=True))):
        for _ in range(warmup):
            ring_attention(q, k, v, None, layout="bhsd", recv_buffers=pool, k_prescaled=kpre, **kw)
        _sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ring_attention(q, k, v, None, layout="bhsd", recv_buffers=pool, k_prescaled=kpre, **kw)
        _sync()
        el = _max_over_ranks(time.perf_counter() - t0) / steps
        flops = 4.0 * S_total * S_total * d * (0.5 if "causal" in kw else 1.0)
        out[name] = {"ms": el * 1e3, "tokens_per_s": S_total / el, "tflops_total": flops / el / 1e12}
    out["exchange_payload_MiB_per_step_per_rank"] = 2 * Sl * d * 2 / 2 ** 20
    out["config"] = (f"sp={N} seq={S_total} d={d} h={H} B=1 attention core, q/k/v shards [1,{H},{Sl},{D}] resident per rank "
                     f"(reference sequence_parallel.py:519-585 ring_exchange call site)")
    return out
