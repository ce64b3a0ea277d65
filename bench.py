#!/usr/bin/env python3
"""Headline benchmark: forward tokens/s of the GPT-2-shaped stack (d=1024, H=16, L=24, S=4096, B=8 per GPU,
bf16, causal FA3 + FusedMLP) on N MI355X, plus the roofline of the dominant kernel and the CPU baseline.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one forward of the whole stack over one synthetic batch resident in HBM.  tokens/s =
B*S / avg latency (reference benchmarks/runners.py:356-358).  N > 1: one process per GPU, the batch
dimension is sharded (independent sequences, no data-path collective) -> weak scaling.  With --parallel-extras
the tensor-parallel (config 3) and ring-attention (config 4) exchange paths are also measured after the timed
region and reported under "extra" (strong-scaling measurements of the same global work); they are opt-in because
a collective that hangs there would take the headline line down with it.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0


def _events_ms(fn, iters, warmup=3, sustain_ms=200.0):
    """Average duration of fn() in ms, HIP events on the stream the kernels are launched on
    (ops.* launch on torch's current stream).  Launches are queued back to back for `sustain_ms` before and
    during the timed region so the chip sits at the sustained clock / power point it holds inside the model."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    n = int(sustain_ms / max(s.elapsed_time(e), 1e-3))
    for _ in range(n):
        fn()
    iters = max(iters, n)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def kernel_rooflines(model, x, iters=10):
    """Per-kernel time of ONE layer's launches on the layer's real weights and activations (layer 0 of the
    benchmark stack, so the operand statistics -- and with them the clock the chip holds -- are the model's own),
    HIP events on the launch stream -> which kernel dominates and its achieved fraction of the MFMA roofline
    (algorithmic FLOPs: SURVEY.md 8d / BASELINE.md section 4)."""
    from mio import ops

    blk = model.h[0]
    B, S, d = x.shape
    H = blk.attn.num_attention_heads
    D = d // H
    M = B * S
    wqkv, bqkv = blk.attn.qkv_proj.weight, blk.attn.qkv_proj.bias
    wo, bo = blk.attn.o_proj.weight, blk.attn.o_proj.bias
    w1, b1 = blk.mlp.mlp.fc1.weight, blk.mlp.mlp.fc1.bias
    w2, b2 = blk.mlp.mlp.fc2.weight, blk.mlp.mlp.fc2.bias
    I = w1.shape[0]
    # weights in the blocked layout, as the modules hand them over at this size (mio/_nn.py linear)
    wqkv_b, wo_b, w1_b, w2_b = (ops.block_weight(t) for t in (wqkv, wo, w1, w2))
    ln1 = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias)
    qkv = ops.gemm_bias_act(ln1, wqkv, bqkv)
    q = qkv[:, :, :d].view(B, S, H, D)
    k = qkv[:, :, d:2 * d].view(B, S, H, D)
    v = qkv[:, :, 2 * d:].view(B, S, H, D)
    ctx = ops.fa3_fwd(q, k, v, causal=True).view(B, S, d)
    att = ops.gemm_bias_act(ctx, wo, bo, residual=x)
    ln2 = ops.layernorm(att, blk.ln_2.weight, blk.ln_2.bias)
    o3, o1 = torch.empty_like(qkv), torch.empty_like(att)
    out = {}
    t = _events_ms(lambda: ops.fa3_fwd(q, k, v, causal=True), iters)
    out["fa3_fwd3_kernel<bf16,causal>"] = dict(ms=t, launches=1, flops=2.0 * B * S * (S + 1) * d)

    t = _events_ms(lambda: ops.gemm_bias_act(ln1, wqkv, bqkv, out=o3, w_blocked=wqkv_b), iters)
    out["gemm4w16p_kernel<bf16,none>"] = dict(ms=t, launches=1, flops=2.0 * M * d * 3 * d)  # qkv
    # the MLP as the model runs it: fc1 + GELU (persistent kernel) writes the blocked intermediate, fc2 + residual reads it
    t = _events_ms(lambda: ops.fused_mlp(ln2, w1, b1, w2, b2, "gelu", residual=att, fc1_blocked=w1_b, fc2_blocked=w2_b), iters)
    out["fused_mlp: gemm4w16p_kernel<bf16,gelu_tanh> + gemm4w16_kernel<bf16,none>"] = dict(
        ms=t, launches=2, flops=4.0 * M * d * I, combined=True)  # two different kernels: not a roofline candidate
    t = _events_ms(lambda: ops.gemm_bias_act(ctx, wo, bo, residual=x, out=o1, w_blocked=wo_b), iters)
    out["gemm4w16_kernel<bf16,none> (out-proj)"] = dict(ms=t, launches=1, flops=2.0 * M * d * d)
    t = _events_ms(lambda: ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias), iters)
    out["layernorm_kernel<bf16>"] = dict(ms=2 * t, launches=2, bytes=2 * 2.0 * M * d * 2)
    return out


def cpu_baseline(d, H, S):
    """Reference CPU path (baseline/inference.py BasicInferenceRunner semantics, oracle port) on this host's
    cores, on a bounded sample: 2 layers, B=1 of the same S/d/H; reported as full-stack-equivalent tokens/s."""
    from oracle.baseline_runner import time_cpu_baseline

    layers = 2
    r = time_cpu_baseline(hidden_size=d, num_heads=H, num_layers=layers, batch=1, seq_len=S, warmup=1, iters=2)
    return r, layers


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seq", type=int, default=4096)
    ap.add_argument("--hidden", type=int, default=1024)
    ap.add_argument("--heads", type=int, default=16)
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--no-extra", action="store_true", help="skip the roofline / cpu-baseline legs (N = 1)")
    ap.add_argument("--parallel-extras", action="store_true",
                    help="N > 1: also time the tensor-parallel and ring-attention paths (RCCL collectives)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIO_BENCH_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N > 1 plumbing
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(0)
    N = world
    dev, dt = "cuda", torch.bfloat16
    B, S, d, H, L = a.batch, a.seq, a.hidden, a.heads, a.layers
    I = 4 * d

    from mio.synthetic import GPT2ShapedStack

    model = GPT2ShapedStack(d, H, L, I, causal=True, precision="bf16", seed=0).to(device=dev, dtype=dt).eval()
    torch.manual_seed(1234 + rank)
    x = torch.randn(B, S, d, device=dev, dtype=dt)  # synthetic hidden states N(0,1), resident in HBM

    def sync_all():
        torch.cuda.synchronize()
        if N > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(a.warmup):
            model(x)
        sync_all()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            model(x)
        sync_all()
        elapsed = time.perf_counter() - t0
    if N > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / a.steps * 1e3
    tokens_per_s = N * B * S * a.steps / elapsed
    flops_step = L * (2.0 * B * S * (S + 1) * d + 8.0 * B * S * d * d + 4.0 * B * S * d * I)

    res = {
        "metric": "forward tokens/sec GPT-2 d=1024 seq=4096 at 1/2/4/8 MI355X; % MFMA roofline",
        "value": tokens_per_s,
        "unit": "tokens/s",
        "n_gpus": N,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "synthetic (random-init weights N(0,0.02), hidden states N(0,1), seed 0)",
        "config": {"workload": f"GPT-2-shaped block stack d={d} h={H} L={L} seq={S} B={B}/GPU causal FA3+FusedMLP(gelu) forward",
                   "global_batch": B * N, "seq_len": S, "parallelism": f"dp{N}" if N > 1 else "single"},
        "model_tflops_per_gpu": flops_step / (ms_per_step * 1e-3) / 1e12,
        "mfma_roofline_frac_end_to_end": flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
    }

    if not a.no_extra and rank == 0 and N == 1:
        try:
            ks = kernel_rooflines(model, x)
            dom = max((k for k in ks if "flops" in ks[k] and not ks[k].get("combined")),
                      key=lambda k: ks[k]["ms"] / ks[k]["launches"])
            kd = ks[dom]
            per_launch_ms = kd["ms"] / kd["launches"]
            ach = kd["flops"] / kd["launches"] / (per_launch_ms * 1e-3) / 1e12
            traffic = None
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):
                try:
                    traffic = json.load(open(tfile)).get(dom)
                except Exception:
                    traffic = None
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                               "avg_launch_ms": per_launch_ms, "flops_per_launch": kd["flops"] / kd["launches"]}
            res["kernels_per_layer"] = {
                k: ({"ms": round(v["ms"], 4), "launches": v["launches"],
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} if "flops" in v else
                    {"ms": round(v["ms"], 4), "launches": v["launches"],
                     "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}) for k, v in ks.items()}
        except Exception as ex:  # the headline number must still be printed
            res["roofline_error"] = repr(ex)
        try:
            r, layers = cpu_baseline(d, H, S)
            res["cpu_baseline"] = {"value": r["tokens_per_s"] * layers / L, "unit": "tokens/s", "cores": r["threads"],
                                   "kind": "port",
                                   "sample": f"oracle/baseline_runner.py (BasicInferenceRunner port) fp32, {layers} of {L} layers, "
                                             f"B=1 S={S} d={d} h={H}, 1 warm-up + 2 timed forwards; value scaled by {layers}/{L} to the full stack"}
        except Exception as ex:
            res["cpu_baseline_error"] = repr(ex)

    if a.parallel_extras and N > 1:
        extra = {}
        try:
            from tools.bench_parallel import bench_tp, bench_ring
            extra["tensor_parallel"] = bench_tp(N, B, S, d, H, I, L, dt, steps=max(2, a.steps // 2))
            extra["ring_attention"] = bench_ring(N, 65536, d, H, dt, steps=3)
        except Exception as ex:
            extra["error"] = repr(ex)
        res["extra"] = extra

    if rank == 0:
        print(json.dumps(res))
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
