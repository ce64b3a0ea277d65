#!/usr/bin/env python3
"""Headline benchmark: forward tokens/s of the GPT-2-shaped stack (d=1024, H=16, L=24, S=4096, B=8 per GPU,
bf16, causal FA3 + FusedMLP) on N MI355X, plus the roofline of the dominant kernel and the CPU baseline.

    python bench.py --gpus N --steps K --warmup W
        N = 1: runs in this process.  N > 1 without a launcher (WORLD_SIZE unset): this process touches no GPU, starts
        N child processes (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and relays rank 0's line.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W          (the driver's form: same code path in every rank)

One "step" = one forward of the whole stack over one synthetic batch resident in HBM.  tokens/s =
B*S / avg latency (reference benchmarks/runners.py:356-358).  N > 1: one process per GPU over RCCL, the batch
dimension is sharded (independent sequences, no data-path collective) -> weak scaling; `ranks_seen` is an all-reduce
of ones over the job.  After the timed region, under "extra" (never inside it, never part of `value`):
  N = 1: paged decode (HBM roofline), the C5 workload (non-causal cross attention d 1280 / Dh 80 + FusedMLP-GELU);
  N > 1: BASELINE configs 3 and 4 -- the tensor-parallel stack (tp 2 and 4: RCCL all-reduce over xGMI, overlapped
         with the row-parallel GEMM vs not) and ring attention at S 65 536 over the N ranks (K/V exchange overlapped
         with the attention kernel vs not).  Each leg is fenced by try/except and a watchdog: a collective that hangs
         costs the leg, not the headline line.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
METRIC = "forward tokens/sec GPT-2 d=1024 seq=4096 at 1/2/4/8 MI355X; % MFMA roofline"


# ----------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` (N > 1, no launcher) -> N ranks
# ----------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n: int) -> int:
    """Parent of a self-launched job: never initialises the GPU; starts one fresh child per rank, relays rank 0's
    stdout (the JSON line), sends the other ranks' stdout to stderr, returns the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        out = None if r == 0 else sys.stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    rc = 0
    try:
        for p in procs:
            p.wait()
            rc = max(rc, abs(p.returncode))
            if p.returncode != 0:  # one rank failed: the others would wait in a collective forever
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------
def _events_ms(fn, iters, warmup=3, sustain_ms=200.0):
    """Average duration of fn() in ms, HIP events on the stream the kernels are launched on
    (ops.* launch on torch's current stream).  Launches are queued back to back for `sustain_ms` before and
    during the timed region so the chip sits at the sustained clock / power point it holds inside the model."""
    import torch

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
    """Per-kernel time of ONE layer's launches on the layer's real weights and activations (layer 1 of the benchmark stack,
    reached through layer 0, so the operand statistics -- and with them the clock the chip holds -- are the model's own),
    HIP events on the launch stream -> which kernel dominates and its achieved fraction of the MFMA roofline
    (algorithmic FLOPs: SURVEY.md 8d / BASELINE.md section 4).  Where the blocks run with their LayerNorms folded into the
    GEMMs (mio._nn.ResidualStream) the launches timed are those: QKV / fc1 normalise in their read-out, out-proj / fc2 write the
    blocked stream + its row statistics, and no LayerNorm kernel runs between them (only in front of block 0 and behind the
    last block)."""
    import torch
    from mio import ops
    from mio._nn import ResidualStream

    B, S, d = x.shape
    blk0, blk = model.h[0], model.h[1 if len(model.h) > 1 else 0]
    H = blk.attn.num_attention_heads
    D = d // H
    M = B * S
    dt = x.dtype
    wo, bo = blk.attn.o_proj.weight, blk.attn.o_proj.bias
    w1, b1 = blk.mlp.mlp.fc1.weight, blk.mlp.mlp.fc1.bias
    w2, b2 = blk.mlp.mlp.fc2.weight, blk.mlp.mlp.fc2.bias
    wqkv, bqkv = blk.attn.qkv_proj.weight, blk.attn.qkv_proj.bias
    I = w1.shape[0]
    fold = (not getattr(model, "no_ln_fold", False)) and len(model.h) > 1 and all(b_.stream_ok(B, S, dt) for b_ in model.h)
    out = {}
    kpre = (D <= 64 and d % 128 == 0 and ops.col_scale_ok(M, 3 * d, d) and ops.fa3_k_prescaled_ok(B, S, S, H, D, 3 * d, 3 * d))
    cs = (d, 2 * d, 1.4426950408889634 / D ** 0.5) if kpre else None
    if fold:
        with torch.no_grad():
            s1 = blk0(x, stream_out=True)  # the residual stream as block 1 receives it: blocked + row statistics
        assert isinstance(s1, ResidualStream)
        ac, mc = blk.attn._cast, blk.mlp.mlp._cast
        wq_f, bq_f = ac.get_ln_folded(blk.attn.qkv_proj, blk.ln_1, dt)
        w1_f, b1_f = mc.get_ln_folded(blk.mlp.mlp.fc1, blk.ln_2, dt)
        wo_b, w2_b = ops.block_weight(wo), ops.block_weight(w2)
        qkv, _ = ops.gemm_ln(s1.blocked, wq_f, bq_f, M=M, N=3 * d, K=d, x_blocked=True, ln_stats=s1.stats, eps=blk.ln_1.eps, col_scale=cs)
        qkv = qkv.view(B, S, 3 * d)
        q, k, v = (qkv[:, :, i * d:(i + 1) * d].view(B, S, H, D) for i in range(3))
        ctx = ops.fa3_fwd(q, k, v, causal=True, k_prescaled=True, out_blocked=True)
        a_b, a_st = ops.gemm_ln(ctx, wo_b, bo, M=M, N=d, K=d, x_blocked=True, residual=s1.blocked, res_blocked=True,
                                out_blocked=True, stats_out=True)
        t = _events_ms(lambda: ops.fa3_fwd(q, k, v, causal=True, k_prescaled=True, out_blocked=True), iters)
        out["fa3_fwd5_kernel<bf16,causal>"] = dict(ms=t, launches=1, flops=2.0 * B * S * (S + 1) * d)
        t = _events_ms(lambda: ops.gemm_ln(s1.blocked, wq_f, bq_f, M=M, N=3 * d, K=d, x_blocked=True, ln_stats=s1.stats,
                                           eps=blk.ln_1.eps, col_scale=cs), iters)
        out["gemm8w_kernel<bf16,none,ln-fold> (qkv)"] = dict(ms=t, launches=1, flops=2.0 * M * d * 3 * d)

        def mlp():
            h, _ = ops.gemm_ln(a_b, w1_f, b1_f, M=M, N=I, K=d, activation="gelu", x_blocked=True, out_blocked=True,
                               ln_stats=a_st, eps=blk.ln_2.eps)
            ops.gemm_ln(h, w2_b, b2, M=M, N=d, K=I, x_blocked=True, residual=a_b, res_blocked=True, out_blocked=True, stats_out=True)

        t = _events_ms(mlp, iters)
        out["fused_mlp: gemm8w_kernel<bf16,gelu_tanh,ln-fold> + gemm8w_kernel<bf16,none,residual,ln-stats>"] = dict(
            ms=t, launches=2, flops=4.0 * M * d * I, combined=True)  # two different kernels: not a roofline candidate
        t = _events_ms(lambda: ops.gemm_ln(ctx, wo_b, bo, M=M, N=d, K=d, x_blocked=True, residual=s1.blocked, res_blocked=True,
                                           out_blocked=True, stats_out=True), iters)
        out["gemm8w_kernel<bf16,none,residual,ln-stats> (out-proj)"] = dict(ms=t, launches=1, flops=2.0 * M * d * d)
        t = _events_ms(lambda: ops.layernorm(x, blk0.ln_1.weight, blk0.ln_1.bias), iters)
        # two LayerNorm launches per FORWARD are left (block 0's ln_1, ln_f): their per-layer share
        out["layernorm_kernel<bf16> (2 launches per forward / L)"] = dict(ms=2 * t / len(model.h), launches=2,
                                                                         bytes=2 * 2.0 * M * d * 2 / len(model.h))
        return out
    # weights in the blocked layout, as the modules hand them over at this size (mio/_nn.py linear)
    wqkv_b, wo_b, w1_b, w2_b = (ops.block_weight(t) for t in (wqkv, wo, w1, w2))
    ln1 = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias)
    # as FlashSelfAttention runs it: the QKV projection's epilogue scales the K columns by softmax_scale * log2(e) in fp32
    # (one rounding) and the attention kernel is told so (k_prescaled) where both kernels support it
    qkv = ops.gemm_bias_act(ln1, wqkv, bqkv, w_blocked=wqkv_b, col_scale=cs)
    q = qkv[:, :, :d].view(B, S, H, D)
    k = qkv[:, :, d:2 * d].view(B, S, H, D)
    v = qkv[:, :, 2 * d:].view(B, S, H, D)
    # ... and where the output projection runs a 256-tile kernel the attention epilogue writes its result in that GEMM's blocked
    # activation layout (ops.fa3_fwd out_blocked), again as FlashSelfAttention does
    oblk = bool(kpre and ops.blocked_weight_ok(M, d, d) and ops.fa3_o_blocked_ok(B, S, S, H, D, 3 * d, 3 * d))
    if oblk:
        ctx = ops.fa3_fwd(q, k, v, causal=True, k_prescaled=True, out_blocked=True)
        att = ops.gemm_bias_act(ctx, wo, bo, residual=x, w_blocked=wo_b, x_blocked_shape=(B, S, d))
    else:
        ctx = ops.fa3_fwd(q, k, v, causal=True, k_prescaled=kpre).view(B, S, d)
        att = ops.gemm_bias_act(ctx, wo, bo, residual=x)
    ln2 = ops.layernorm(att, blk.ln_2.weight, blk.ln_2.bias)
    o3, o1 = torch.empty_like(qkv), torch.empty_like(att)
    t = _events_ms(lambda: ops.fa3_fwd(q, k, v, causal=True, k_prescaled=kpre, out_blocked=oblk), iters)
    out["fa3_fwd5_kernel<bf16,causal>" if kpre else "fa3_fwd4_kernel<bf16,causal>"] = dict(ms=t, launches=1, flops=2.0 * B * S * (S + 1) * d)

    t = _events_ms(lambda: ops.gemm_bias_act(ln1, wqkv, bqkv, out=o3, w_blocked=wqkv_b, col_scale=cs), iters)
    out["gemm8w_kernel<bf16,none> (qkv)"] = dict(ms=t, launches=1, flops=2.0 * M * d * 3 * d)
    # the MLP as the model runs it: fc1 + GELU writes the blocked intermediate, fc2 + residual reads it
    t = _events_ms(lambda: ops.fused_mlp(ln2, w1, b1, w2, b2, "gelu", residual=att, fc1_blocked=w1_b, fc2_blocked=w2_b), iters)
    out["fused_mlp: gemm8w_kernel<bf16,gelu_tanh> + gemm8w_kernel<bf16,none,residual>"] = dict(
        ms=t, launches=2, flops=4.0 * M * d * I, combined=True)  # two different kernels: not a roofline candidate
    t = _events_ms(lambda: ops.gemm_bias_act(ctx, wo, bo, residual=x, out=o1, w_blocked=wo_b,
                                             x_blocked_shape=(B, S, d) if oblk else None), iters)
    out["gemm8w_kernel<bf16,none,residual> (out-proj)"] = dict(ms=t, launches=1, flops=2.0 * M * d * d)
    t = _events_ms(lambda: ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias), iters)
    out["layernorm_kernel<bf16>"] = dict(ms=2 * t, launches=2, bytes=2 * 2.0 * M * d * 2)
    return out


def _decode_traffic(kernel_substr):
    """HBM fetch bytes per launch of a decode kernel from the committed PMC pass (profiles/r03_pmc_decode.json, tools/decode_pmc.py)."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_decode.json")))
        for k, v in d.items():
            if kernel_substr in k:
                return v["fetch_bytes_per_launch"]
    except Exception:
        pass
    return None


def decode_leg(dt):
    """Paged decode step: B 64, H 16, D 64, context 4096, block 16 -- 1 GiB of K/V cache, streams from
    HBM.  Bound: HBM.  Algorithmic bytes = 2 * B * ctx * Hkv * D * 2 (every cached K and V element once)."""
    import torch
    from mio import ops

    Bd, Hd, Dd, bs, ctx = 64, 16, 64, 16, 4096
    nblk = Bd * ctx // bs
    kc = torch.randn(nblk, 1, bs, Hd, Dd, device="cuda", dtype=dt)
    vc = torch.randn(nblk, 1, bs, Hd, Dd, device="cuda", dtype=dt)
    bt = torch.randperm(nblk, device="cuda").view(Bd, -1).to(torch.int32)
    cl = torch.full((Bd,), ctx, device="cuda", dtype=torch.int32)
    q = torch.randn(Bd, Hd, 1, Dd, device="cuda", dtype=dt)
    o = torch.empty_like(q)
    ms = _events_ms(lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0), 20, sustain_ms=100.0)
    nbytes = 2.0 * Bd * ctx * Hd * Dd * 2
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"workload": f"paged decode q_len 1, B {Bd} H {Hd} D {Dd} ctx {ctx} block {bs} (random physical blocks)",
            "kernel": "decode_rows_kernel + decode_reduce_kernel", "bound": "hbm", "ms": ms, "achieved": gbs,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes,
            "traffic": _decode_traffic("decode_rows_kernel"),
            "tokens_per_s": Bd / (ms * 1e-3)}


def decode_gqa_leg(dt):
    """Paged decode with grouped queries (8 query heads per kv head: H 32, Hkv 4, D 128, B 64, ctx 4096 -- 512 MiB of cache):
    decode_gqa_kernel, the q . K^T / P . V of the 8 (padded to 16) query vectors on the matrix core.  Bound: HBM."""
    import torch
    from mio import ops

    Bd, Hd, Hkv, Dd, bs, ctx = 64, 32, 4, 128, 16, 4096
    nblk = Bd * ctx // bs
    kc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    vc = torch.randn(nblk, 1, bs, Hkv, Dd, device="cuda", dtype=dt)
    bt = torch.randperm(nblk, device="cuda").view(Bd, -1).to(torch.int32)
    cl = torch.full((Bd,), ctx, device="cuda", dtype=torch.int32)
    q = torch.randn(Bd, Hd, 1, Dd, device="cuda", dtype=dt)
    o = torch.empty_like(q)
    ms = _events_ms(lambda: ops.paged_attention_forward(q, o, kc, vc, bt, cl, bs, ctx, 0), 20, sustain_ms=100.0)
    nbytes = 2.0 * Bd * ctx * Hkv * Dd * 2
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"workload": f"paged decode q_len 1, B {Bd} H {Hd} Hkv {Hkv} D {Dd} ctx {ctx} block {bs} (random physical blocks)",
            "kernel": "decode_gqa_kernel + decode_reduce_kernel", "bound": "hbm", "ms": ms, "achieved": gbs,
            "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "bytes_per_launch": nbytes,
            "traffic": _decode_traffic("decode_gqa_kernel"),
            "tokens_per_s": Bd / (ms * 1e-3)}


def attention_functional_leg(dt):
    """The reference's functional entry point, triton_flash_attention(q, k, v, causal=True) (flash_attention_kernels.py:1150-1358),
    at the C2 attention shape with plain (not pre-scaled) K: fa3_fwd5_kernel's KPRE = false form."""
    import torch
    from mio import ops

    B, S, H, D = 8, 4096, 16, 64
    torch.manual_seed(5)
    q, k, v = (torch.randn(B, S, H, D, device="cuda", dtype=dt) for _ in range(3))
    out = {}
    for causal in (True, False):
        ms = _events_ms(lambda: ops.flash_attention(q, k, v, causal=causal), 10)
        flops = (2.0 * B * S * (S + 1) if causal else 4.0 * B * S * S) * H * D
        out["causal" if causal else "non_causal"] = {"ms": ms, "tflops": flops / (ms * 1e-3) / 1e12,
                                                     "mfma_roofline_frac": flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}
    out["workload"] = f"ops.flash_attention(q, k, v) B={B} S={S} H={H} D={D}, K not pre-scaled"
    return out


def swiglu_leg(dt):
    """FusedMLP-SwiGLU at the C2 shape (M 32768, d 1024, I 4096; 6 M d I FLOPs): the gated form of gemm8w_kernel on the
    interleaved gate / up blocked weight + fc2 with bias and residual (reference mlp_kernels.py:417-641)."""
    import torch
    from mio import ops

    B, S, d, I = 8, 4096, 1024, 4096
    M = B * S
    torch.manual_seed(11)
    x = torch.randn(B, S, d, device="cuda", dtype=dt)
    r = torch.randn(B, S, d, device="cuda", dtype=dt)
    wu, wg = ((torch.randn(I, d, device="cuda") * 0.02).to(dt) for _ in range(2))
    w2 = (torch.randn(d, I, device="cuda") * 0.02).to(dt)
    bu, bg = ((torch.randn(I, device="cuda") * 0.02).to(dt) for _ in range(2))
    b2 = (torch.randn(d, device="cuda") * 0.02).to(dt)
    blocked = ops.fused_mlp_blocked_weight_ok(M, d, I, "swiglu")
    kw = dict(fc1_blocked=ops.block_weight_glu(wg, wu), fc2_blocked=ops.block_weight(w2)) if blocked else {}
    ms = _events_ms(lambda: ops.fused_mlp(x, wu, bu, w2, b2, "swiglu", wg, bg, residual=r, **kw), 10)
    flops = 6.0 * M * d * I
    return {"workload": f"FusedMLP swiglu M={M} d={d} I={I} + bias + residual (two launches)", "blocked_weights": bool(blocked),
            "kernel": "gemm8w_kernel<bf16,swiglu> + gemm8w_kernel<bf16,none,residual>", "ms": ms,
            "tflops": flops / (ms * 1e-3) / 1e12, "mfma_roofline_frac": flops / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS}


def c5_leg(dt, steps=3):
    """BASELINE configs[4]: diffusion-style non-causal cross attention d 1280 (H 16 -> Dh 80), Sq = Sk = 4096, +
    FusedMLP-GELU I 5120; 4 blocks, B 8, context from a separate [B, Sk, d] tensor."""
    import torch
    from mio.synthetic import CrossAttentionStack

    B, S, d, H, L = 8, 4096, 1280, 16, 4
    I = 4 * d
    model = CrossAttentionStack(d, H, L, I, "bf16", seed=0).to(device="cuda", dtype=dt).eval()
    torch.manual_seed(7)
    x = torch.randn(B, S, d, device="cuda", dtype=dt)
    ctx = torch.randn(B, S, d, device="cuda", dtype=dt)
    with torch.no_grad():
        for _ in range(4):  # first calls: weight repack / fold, hipFuncSetAttribute, (under rocprofv3) the tracer's first-launch work
            model(x, ctx)
        torch.cuda.synchronize()
        # groups of `steps` queued forwards, the median group: one allocator or driver hiccup in a 25-ms window (a run right behind
        # the decode legs' 2 GiB of freed caches showed 14.5 ms once) does not become the number
        groups = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(steps):
                model(x, ctx)
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / steps)
        el = sorted(groups)[1]
    flops = L * (4.0 * B * S * S * d + 8.0 * B * S * d * d + 4.0 * B * S * d * I)
    del model
    torch.cuda.empty_cache()
    return {"workload": f"cross-attention blocks d={d} h={H} (Dh 80) Sq=Sk={S} B={B} L={L} non-causal + FusedMLP(gelu) I={I}",
            "ms_per_step": el * 1e3, "ms_per_step_groups": [round(g_ * 1e3, 3) for g_ in groups], "tokens_per_s": B * S / el,
            "model_tflops": flops / el / 1e12,
            "mfma_roofline_frac": flops / el / 1e12 / PEAK_BF16_TFLOPS}


def cpu_baseline(d, H, S):
    """Reference CPU path (baseline/inference.py BasicInferenceRunner semantics, oracle port) on this host's
    cores, on a bounded sample: 2 layers, B=1 of the same S/d/H; reported as full-stack-equivalent tokens/s."""
    from oracle.baseline_runner import time_cpu_baseline

    layers = 2
    r = time_cpu_baseline(hidden_size=d, num_heads=H, num_layers=layers, batch=1, seq_len=S, warmup=1, iters=2)
    return r, layers


def cpu_baseline_c1():
    """BASELINE configs[0]: GPT-2-small shape (d 768, h 12, L 12) seq 128 B 1 through the BasicInferenceRunner port."""
    from oracle.baseline_runner import time_cpu_baseline

    r = time_cpu_baseline(hidden_size=768, num_heads=12, num_layers=12, batch=1, seq_len=128, warmup=2, iters=5)
    return {"value": r["tokens_per_s"], "unit": "tokens/s", "cores": r["threads"], "kind": "port",
            "sample": "GPT-2-small-shaped block stack d=768 h=12 L=12, B=1 S=128, fp32, 2 warm-up + 5 timed forwards "
                      "(hidden states in / out: no embedding or LM head)", "avg_latency_ms": r["avg_latency_s"] * 1e3}


class Watchdog:
    """If a fenced leg does not finish in `seconds`, rank 0 prints the line it has (with the leg marked) and every rank
    leaves: a hung collective cannot be cancelled from Python."""

    def __init__(self, rank, res):
        self.rank, self.res, self.timer, self.leg, self.printed = rank, res, None, None, False

    def arm(self, leg, seconds):
        self.disarm()
        self.leg = leg
        self.timer = threading.Timer(seconds, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None

    def _fire(self):
        if self.rank == 0 and not self.printed:
            self.res.setdefault("extra", {})[self.leg] = {"error": "timed out (collective did not complete)"}
            sys.stdout.write(json.dumps(self.res) + "\n")
            sys.stdout.flush()
        os._exit(0 if self.rank == 0 else 3)


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
    ap.add_argument("--no-extra", action="store_true", help="only the headline line: no roofline / cpu / extra legs")
    ap.add_argument("--ring-seq", type=int, default=65536, help="N > 1: total sequence length of the ring-attention leg")
    a = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
    if world > 1:
        import datetime
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("MIO_BENCH_BACKEND", "nccl")  # "gloo": single-GPU rehearsal of the N > 1 plumbing
        tmo = datetime.timedelta(seconds=300)
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, timeout=tmo,
                                    device_id=torch.device("cuda", local_rank))
        else:
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    else:
        torch.cuda.set_device(0)
    N = world
    dev, dt = "cuda", torch.bfloat16
    B, S, d, H, L = a.batch, a.seq, a.hidden, a.heads, a.layers
    I = 4 * d
    cdev = dev if backend in (None, "nccl") else "cpu"  # where small control tensors of a collective live

    ranks_seen = 1
    if N > 1:
        one = torch.ones(1, device=cdev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

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
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / a.steps * 1e3
    tokens_per_s = N * B * S * a.steps / elapsed
    flops_step = L * (2.0 * B * S * (S + 1) * d + 8.0 * B * S * d * d + 4.0 * B * S * d * I)

    res = {
        "metric": METRIC,
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
        "ranks_seen": ranks_seen,
        "backend": backend or "none",
        "model_tflops_per_gpu": flops_step / (ms_per_step * 1e-3) / 1e12,
        "mfma_roofline_frac_end_to_end": flops_step / (ms_per_step * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
    }
    dog = Watchdog(rank, res)

    if not a.no_extra and rank == 0 and N == 1:
        try:
            ks = kernel_rooflines(model, x)
            dom = max((k for k in ks if "flops" in ks[k] and not ks[k].get("combined")),
                      key=lambda k: ks[k]["ms"] / ks[k]["launches"])
            kd = ks[dom]
            per_launch_ms = kd["ms"] / kd["launches"]
            ach = kd["flops"] / kd["launches"] / (per_launch_ms * 1e-3) / 1e12
            traffic, tsrc = None, None
            tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(tfile):
                try:
                    traffic = json.load(open(tfile)).get(dom)
                    tsrc = "profiles/pmc_traffic.json (separate rocprofv3 --pmc passes of this command, committed; " \
                           "not re-measured in this run)"
                except Exception:
                    traffic = None
            res["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": ach, "peak": PEAK_BF16_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach / PEAK_BF16_TFLOPS, "traffic": traffic,
                               "traffic_source": tsrc,
                               "avg_launch_ms": per_launch_ms, "flops_per_launch": kd["flops"] / kd["launches"]}
            res["kernels_per_layer"] = {
                k: ({"ms": round(v["ms"], 4), "launches": v["launches"],
                     "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1)} if "flops" in v else
                    {"ms": round(v["ms"], 4), "launches": v["launches"],
                     "GBps": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)}) for k, v in ks.items()}
        except Exception as ex:  # the headline number must still be printed
            res["roofline_error"] = repr(ex)
        del model
        torch.cuda.empty_cache()
        extra = res.setdefault("extra", {})
        for name, leg in (("decode_roofline", lambda: decode_leg(dt)), ("decode_gqa_roofline", lambda: decode_gqa_leg(dt)),
                          ("c5", lambda: c5_leg(dt)), ("swiglu", lambda: swiglu_leg(dt)),
                          ("attention_functional", lambda: attention_functional_leg(dt))):
            try:
                extra[name] = leg()
            except Exception as ex:
                extra[name] = {"error": repr(ex)}
        try:
            r, layers = cpu_baseline(d, H, S)
            res["cpu_baseline"] = {"value": r["tokens_per_s"] * layers / L, "unit": "tokens/s", "cores": r["threads"],
                                   "kind": "port",
                                   "sample": f"oracle/baseline_runner.py (BasicInferenceRunner port) fp32, {layers} of {L} layers, "
                                             f"B=1 S={S} d={d} h={H}, 1 warm-up + 2 timed forwards; value scaled by {layers}/{L} to the full stack"}
        except Exception as ex:
            res["cpu_baseline_error"] = repr(ex)
        try:
            res["cpu_baseline_c1"] = cpu_baseline_c1()
        except Exception as ex:
            res["cpu_baseline_c1_error"] = repr(ex)

    if N > 1 and not a.no_extra:
        # BASELINE configs 3 and 4 (strong-scaling measurements of fixed global work; never part of `value`)
        del model
        torch.cuda.empty_cache()
        from tools.bench_parallel import bench_ring, bench_tp
        extra = res.setdefault("extra", {})
        legs = [(f"tensor_parallel_tp{tp}", 240, (lambda tp=tp: bench_tp(N, tp, B, S, d, H, I, L, dt, steps=max(2, a.steps // 2))))
                for tp in (2, 4) if N % tp == 0]
        legs.append((f"ring_attention_sp{N}", 420, lambda: bench_ring(N, a.ring_seq, d, H, dt, steps=1 if N < 4 else 2)))
        try:  # what these legs should show, from the single-GPU kernel times + link rate (tools/scale_model.py)
            from tools.scale_model import predict
            model_pred = predict(L=L, B=B, S=S, d=d, H=H, I=I, ring_S=a.ring_seq, measured_single_ms=res["ms_per_step"])
        except Exception:
            model_pred = {}
        res["predicted"] = {k: model_pred[k] for k in (f"dp{N}",) if k in model_pred}
        for name, budget_s, leg in legs:
            dog.arm(name, budget_s)
            try:
                extra[name] = leg()
            except Exception as ex:
                extra[name] = {"error": repr(ex)}
            pk = name if name in model_pred else ("ring_attention_sp8" if name.startswith("ring_attention") and N == 8 else None)
            if pk and isinstance(extra[name], dict):
                extra[name]["predicted"] = model_pred[pk]
            dog.disarm()

    if rank == 0:
        print(json.dumps(res))
        sys.stdout.flush()
        dog.printed = True
    if N > 1:
        dog.arm("shutdown", 60)
        dist.barrier()
        dist.destroy_process_group()
        dog.disarm()


if __name__ == "__main__":
    main()
