"""The distributed wrappers with the REAL HIP kernels: two ranks sharing cuda:0 over gloo (a one-GPU box cannot
host two RCCL ranks; gloo moves the same tensors through the host).  What is checked is that the kernels, the
(o, lse) carry, the strided views and the schedules compose correctly on device memory."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, args):
    for p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0")
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world, *args)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _worker_rccl(rank, world, port, fn_name, args):
    """One rank on the real backend ("nccl" = RCCL): all a one-GPU box can host."""
    for p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    torch.cuda.set_device(0)
    torch.set_grad_enabled(False)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        globals()[fn_name](*args)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2, args=()):
    mp.spawn(_worker, args=(world, _free_port(), fn_name, args), nprocs=world, join=True)


def _w_ring(rank, world, exchange, causal, zigzag, layout, kpre=False):
    import math
    import oracle
    from mio.parallelism.sequence_parallel import ring_attention, zigzag_shard
    torch.manual_seed(1)
    B, H, S, D = 1, 4, 512 * world, 64
    dt = torch.float16
    q, k, v = (torch.randn(B, S, H, D).to(dt) for _ in range(3))
    if kpre:  # K as the projection's col_scale epilogue hands it over; the checker works on the same rounded K~ in base 2
        k = (k.float() * (math.log2(math.e) / math.sqrt(D))).to(dt)
        ref = oracle.attention_with_lse(q, k, v, causal=causal, softmax_scale=math.log(2.0))[0]
    else:
        ref = oracle.standard_attention(q, k, v, causal=causal)
    if zigzag:
        loc = [zigzag_shard(t, rank, world, 1) for t in (q, k, v)]
        ref_loc = zigzag_shard(ref, rank, world, 1)
    else:
        n = S // world
        loc = [t[:, rank * n:(rank + 1) * n] for t in (q, k, v)]
        ref_loc = ref[:, rank * n:(rank + 1) * n]
    loc = [(t.permute(0, 2, 1, 3) if layout == "bhsd" else t).contiguous().cuda() for t in loc]
    out = ring_attention(*loc, None, layout=layout, causal=causal, zigzag=zigzag, exchange=exchange, k_prescaled=kpre)
    out = out.cpu()
    if layout == "bhsd":
        out = out.permute(0, 2, 1, 3)
    refq = ref_loc.to(dt).float()
    rel = ((out.float() - refq).abs().mean() / refq.abs().mean()).item()
    assert rel < 1e-3, (exchange, causal, zigzag, layout, rel)


def _w_tp(rank, world):
    import oracle
    from mio.parallelism import TensorParallelConfig, TensorParallelMLP, TensorParallelAttention
    torch.manual_seed(0)
    d, I, H, B, S = 256, 1024, 4, 2, 640
    dt = torch.bfloat16
    cfg = TensorParallelConfig(world_size=world, tp_size=world, overlap_chunks=3)
    x = torch.randn(B, S, d).to(dt)
    res = torch.randn(B, S, d).to(dt)
    w1, b1 = (torch.randn(I, d) * .05).to(dt), (torch.randn(I) * .05).to(dt)
    w2, b2 = (torch.randn(d, I) * .05).to(dt), (torch.randn(d) * .05).to(dt)
    mlp = TensorParallelMLP(d, I, cfg, activation="gelu").to("cuda", dt)
    per = I // world
    mlp.dense_h_to_4h.weight.copy_(w1[rank * per:(rank + 1) * per]); mlp.dense_h_to_4h.bias.copy_(b1[rank * per:(rank + 1) * per])
    mlp.dense_4h_to_h.weight.copy_(w2[:, rank * per:(rank + 1) * per]); mlp.dense_4h_to_h.bias.copy_(b2)
    y = mlp(x.cuda(), residual=res.cuda()).cpu()
    ref = oracle.fused_mlp(x, w1, b1, w2, b2, "gelu", residual=res)
    rel = ((y.double() - ref).abs().mean() / ref.abs().mean()).item()
    assert rel < 6e-3, rel
    att = TensorParallelAttention(d, H, cfg, causal=True).to("cuda", dt)
    ws = [(torch.randn(d, d) * .05).to(dt) for _ in range(4)]
    bs = [(torch.randn(d) * .05).to(dt) for _ in range(4)]
    pd = d // world
    for lin, w, b in zip((att.query, att.key, att.value), ws, bs):
        lin.weight.copy_(w[rank * pd:(rank + 1) * pd]); lin.bias.copy_(b[rank * pd:(rank + 1) * pd])
    att.output.weight.copy_(ws[3][:, rank * pd:(rank + 1) * pd]); att.output.bias.copy_(bs[3])
    y = att(x.cuda()).cpu()
    F = torch.nn.functional
    q, k, v = (F.linear(x.double(), w.double(), b.double()).view(B, S, H, d // H) for w, b in zip(ws[:3], bs[:3]))
    ref = F.linear(oracle.standard_attention(q, k, v, causal=True).reshape(B, S, d), ws[3].double(), bs[3].double())
    rel = ((y.double() - ref).abs().mean() / ref.abs().mean()).item()
    assert rel < 6e-3, rel


def _stack_and_input(seed=0):
    from mio.synthetic import GPT2ShapedStack
    d, H, L, I, B, S = 256, 4, 2, 1024, 2, 1024
    stack = GPT2ShapedStack(d, H, L, I, causal=True, precision="bf16", seed=seed).to("cuda", torch.bfloat16).eval()
    with torch.no_grad():  # non-trivial biases / LayerNorm parameters (the stack initialises them to 0 / 1)
        g = torch.Generator().manual_seed(seed + 1)
        for m in stack.modules():
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.02)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    x = torch.randn(B, S, d, generator=torch.Generator().manual_seed(seed + 2)).to("cuda", torch.bfloat16)
    return stack, x


def _oracle_stack_rows(stack, x, rows):
    """fp64 oracle of the 2-layer GPT-2-shaped stack (causal) for sampled (batch, position) rows: every layer is evaluated on
    the whole batch element up to that position (the next layer's keys need all earlier tokens), test-size only."""
    import oracle
    F = torch.nn.functional
    c = lambda t: t.detach().double().cpu()
    out = []
    for b in sorted({b for b, _ in rows}):
        h = c(x[b:b + 1])
        for blk in stack.h:
            at = blk.attn
            H = at.num_attention_heads
            d = h.shape[-1]
            a = oracle.layernorm(h, c(blk.ln_1.weight), c(blk.ln_1.bias), blk.ln_1.eps)
            qkv = F.linear(a, c(at.qkv_proj.weight), c(at.qkv_proj.bias))
            q, k, v = (t.view(1, -1, H, d // H) for t in qkv.split(d, dim=-1))
            o = oracle.standard_attention(q, k, v, causal=True).reshape(1, -1, d)
            h = F.linear(o, c(at.o_proj.weight), c(at.o_proj.bias)) + h
            m = blk.mlp.mlp
            h = oracle.fused_mlp(oracle.layernorm(h, c(blk.ln_2.weight), c(blk.ln_2.bias), blk.ln_2.eps), c(m.fc1.weight),
                                 c(m.fc1.bias), c(m.fc2.weight), c(m.fc2.bias), "gelu", residual=h)
        h = oracle.layernorm(h, c(stack.ln_f.weight), c(stack.ln_f.bias), stack.ln_f.eps)
        out.append((b, h[0]))
    full = dict(out)
    return torch.stack([full[b][r] for b, r in rows])


def _rel(a, b):
    return ((a.float() - b.float()).abs().mean() / b.float().abs().mean()).item()


def _w_sharded_stack(rank, world, mode, zigzag, exchange):
    """SequenceParallelConverter + SequenceShardedModule (reference sequence_parallel.py:88-342, 723-920) on a 2-layer
    GPT-2-shaped stack with the real kernels: shard [B,S,d] over the ranks, ring / full attention in every layer,
    token-wise LayerNorm + FusedMLP on the shard, gather -- against the unsharded HIP stack on the same weights."""
    from mio.parallelism import SequenceParallelConfig, SequenceParallelConverter, SequenceShardedModule
    from mio.parallelism.sequence_parallel import SequenceParallelAttention
    stack, x = _stack_and_input()
    ref = stack(x)
    cfg = SequenceParallelConfig(world_size=world, sp_size=world, attention_handling=mode, exchange=exchange,
                                 causal=True, zigzag=zigzag)
    sharded = SequenceParallelConverter(cfg).convert_model(stack)
    assert isinstance(sharded, SequenceShardedModule)
    atts = [m for m in sharded.modules() if isinstance(m, SequenceParallelAttention)]
    assert len(atts) == 2 and all(a.config.causal for a in atts)
    y = sharded(x)
    assert y.shape == ref.shape
    rel = _rel(y, ref)
    assert rel < 4e-3, (mode, zigzag, exchange, rel)
    rows = [(0, 0), (0, 511), (1, 512), (1, 1023), (0, 777)]  # both ranks' shards, incl. the shard boundary
    want = _oracle_stack_rows(stack, x, rows)
    got = torch.stack([y[b, r] for b, r in rows]).double().cpu()
    orel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert orel < 6e-3, ("oracle rows", mode, zigzag, exchange, orel)
    if cfg.buffer_reuse and mode == "ring" and exchange == "mesh":
        pools = [a._recv_buffers for a in atts]
        ptrs = [[b.data_ptr() for c in next(iter(p.values())) for b in c] for p in pools]
        sharded(x)  # second forward: the receive buffers are the same allocations
        assert ptrs == [[b.data_ptr() for c in next(iter(p.values())) for b in c] for p in pools]


def _w_parallel_groups(rank, world):
    """initialize_parallel_groups (reference parallel_utils.py:882-1002) with the real kernels: tensor 2 x sequence 1
    (ModelParallelConverter on the registered tensor group) and tensor 1 x sequence 2 (SequenceParallelConverter on the
    registered sequence group), each against the unsharded HIP stack."""
    from mio.parallelism import (ModelParallelConverter, SequenceParallelConfig, SequenceParallelConverter,
                                 TensorParallelAttention, TensorParallelConfig, TensorParallelMLP)
    from mio.parallelism.parallel_utils import (ParallelConfig, get_process_group_for_operation,
                                                initialize_parallel_groups)
    stack, x = _stack_and_input(seed=3)
    ref = stack(x)
    g = initialize_parallel_groups(ParallelConfig(world, tensor_parallel_size=2, sequence_parallel_size=1))
    assert g["tensor"] is not None and g["sequence"] is None and get_process_group_for_operation("tensor") is g["tensor"]
    tcfg = TensorParallelConfig(world_size=world, tp_size=2, overlap_chunks=2)
    assert tcfg.get_tp_group() is g["tensor"] and tcfg.tp_rank() == rank
    tp_model = ModelParallelConverter(tcfg).convert_model(stack)
    assert sum(isinstance(m, TensorParallelAttention) for m in tp_model.modules()) == 2
    assert sum(isinstance(m, TensorParallelMLP) for m in tp_model.modules()) == 2
    y = tp_model(x)
    rel = _rel(y, ref)
    # tensor parallel sums per-rank partial GEMM results that were rounded to bf16 for the wire (reference
    # tensor_parallel.py:299-308 reduces in the activation dtype too): one extra rounding per rank and row-parallel GEMM
    assert rel < 6e-3, ("tp2", rel)
    g = initialize_parallel_groups(ParallelConfig(world, tensor_parallel_size=1, sequence_parallel_size=2))
    assert g["tensor"] is None and g["sequence"] is not None
    scfg = SequenceParallelConfig(world_size=world, sp_size=2, attention_handling="ring", exchange="ring", causal=True)
    assert scfg.get_sp_group() is g["sequence"] and scfg.get_rank_info() == (rank, 0)
    y = SequenceParallelConverter(scfg).convert_model(stack)(x)
    rel = _rel(y, ref)
    assert rel < 4e-3, ("sp2", rel)


def _spy_linear():
    """Record the col_scale argument of every _local.linear call (which branch a module took)."""
    from mio.parallelism import _local
    seen, real = [], _local.linear

    def spy(x, weight, bias=None, activation="none", residual=None, out=None, col_scale=None):
        seen.append(col_scale)
        return real(x, weight, bias, activation, residual, out, col_scale)
    _local.linear = spy
    return seen


def _w_kpre_module_branches(rank, world, which):
    """The k_prescaled branches of the parallel modules with the REAL kernels at shapes where they are taken
    (ops.col_scale_ok: the 256-tile GEMM, hidden 1024 = 16 heads of 64): TensorParallelAttention's fused [3n, hidden]
    projection with col_scale = (n, 2n) + strided q / k / v views + k_prescaled launch, and SequenceParallelAttention's K
    projection with col_scale + ring of pre-scaled K shards with the (o_acc, lse) carry.  Oracle on sampled rows."""
    import oracle
    from mio.parallelism import SequenceParallelAttention, SequenceParallelConfig, TensorParallelAttention, TensorParallelConfig
    F = torch.nn.functional
    d, H, dt = 1024, 16, torch.bfloat16
    D = d // H
    seen = _spy_linear()
    g = torch.Generator().manual_seed(7)
    ws = [(torch.randn(d, d, generator=g) * 0.03).to(dt) for _ in range(4)]
    bs = [(torch.randn(d, generator=g) * 0.03).to(dt) for _ in range(4)]

    def oracle_rows(x, causal, rows):  # [(b, s)] -> fp64 reference of the block output rows
        q, k, v = (F.linear(x.double(), w.double(), b.double()) for w, b in zip(ws[:3], bs[:3]))
        out = []
        for b, r in rows:
            n = r + 1 if causal else x.shape[1]
            o = oracle.standard_attention(q[b:b + 1, r:r + 1].view(1, 1, H, D), k[b:b + 1, :n].view(1, n, H, D),
                                          v[b:b + 1, :n].view(1, n, H, D), causal=False)
            out.append(F.linear(o.reshape(d), ws[3].double(), bs[3].double()))
        return torch.stack(out)

    if which == "sp":
        return _kpre_sp_part(rank, world, d, H, dt, g, ws, bs, seen, oracle_rows)
    # --- tensor parallel, tp = 2: per rank n = 512 columns of each of q / k / v
    B, S = 3, 4096
    x = torch.randn(B, S, d, generator=g).to(dt)
    cfg = TensorParallelConfig(world_size=world, tp_size=world, overlap_chunks=2)
    att = TensorParallelAttention(d, H, cfg, causal=True).to("cuda", dt)
    pd = d // world
    for lin, w, b in zip((att.query, att.key, att.value), ws, bs):
        lin.weight.copy_(w[rank * pd:(rank + 1) * pd]); lin.bias.copy_(b[rank * pd:(rank + 1) * pd])
    att.output.weight.copy_(ws[3][:, rank * pd:(rank + 1) * pd]); att.output.bias.copy_(bs[3])
    seen.clear()
    y = att(x.cuda()).cpu()
    assert any(c is not None and c[0] == pd and c[1] == 2 * pd for c in seen), ("TensorParallelAttention did not pre-scale K", seen)
    rows = [(0, 0), (0, 255), (1, 256), (1, 1777), (2, 4095), (2, 3000)]
    want = oracle_rows(x, True, rows)
    got = torch.stack([y[b, r] for b, r in rows]).double()
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < 6e-3, ("tp kpre", rel)



def _kpre_sp_part(rank, world, d, H, dt, g, ws, bs, seen, oracle_rows):
    from mio.parallelism import SequenceParallelAttention, SequenceParallelConfig
    # --- sequence parallel ring, sp = 2: 4 x 4096 local tokens per rank
    B, S = 4, 8192
    x = torch.randn(B, S, d, generator=g).to(dt)
    scfg = SequenceParallelConfig(world_size=world, sp_size=world, attention_handling="ring", exchange="mesh", causal=True,
                                  zigzag=True)
    sp = SequenceParallelAttention(d, H, scfg, attention_dropout=0.0).to("cuda", dt)
    for lin, w, b in zip((sp.query, sp.key, sp.value, sp.output), ws, bs):
        lin.weight.copy_(w); lin.bias.copy_(b)
    from mio.parallelism.sequence_parallel import zigzag_shard
    xl = zigzag_shard(x, rank, world, 1).contiguous().cuda()
    seen.clear()
    yl = sp(xl).cpu()
    assert any(c is not None and c[0] == 0 and c[1] == d for c in seen), ("SequenceParallelAttention did not pre-scale K", seen)
    # local row r of a zig-zag shard: blocks (rank, 2 world - 1 - rank) of S / (2 world) tokens
    blk = S // (2 * world)
    pos = lambda r: (rank * blk + r) if r < blk else ((2 * world - 1 - rank) * blk + (r - blk))
    loc = [(0, 0), (1, blk - 1), (2, blk), (3, 2 * blk - 1), (0, 1234)]
    want = oracle_rows(x, True, [(b, pos(r)) for b, r in loc])
    got = torch.stack([yl[b, r] for b, r in loc]).double()
    rel = ((got - want).abs().mean() / want.abs().mean()).item()
    assert rel < 4e-3, ("sp ring kpre", rel)


def _w_tp2_x_sp2_hip(rank, world):
    """tensor (2) x sequence (2) on FOUR ranks with the real kernels (reference parallel_utils.py:882-1002): ring attention
    over the sequence group on this rank's heads, row-parallel out-projection summed over the tensor group == the dense
    block output for the rank's tokens (oracle)."""
    import oracle
    from mio.parallelism import RowParallelLinear, SequenceParallelConfig, TensorParallelConfig
    from mio.parallelism.sequence_parallel import ring_attention
    F = torch.nn.functional
    dt = torch.float16
    tcfg = TensorParallelConfig(world_size=world, tp_size=2)
    scfg = SequenceParallelConfig(world_size=world, sp_size=2, tp_size=2, exchange="mesh")
    tg, sg = tcfg.get_tp_group(), scfg.get_sp_group()
    tp_r, sp_r = tcfg.tp_rank(), scfg.get_rank_info()[0]
    assert (tp_r, sp_r) == (rank % 2, rank // 2)
    torch.manual_seed(0)  # the same tensors on every rank
    B, S, H, D = 2, 1024, 4, 64
    d = H * D
    q, k, v = (torch.randn(B, S, H, D).to(dt) for _ in range(3))
    wo, bo = (torch.randn(d, d) * 0.05).to(dt), (torch.randn(d) * 0.05).to(dt)
    dense = F.linear(oracle.standard_attention(q, k, v, causal=True).reshape(B, S, d), wo.double(), bo.double())
    hs = slice(tp_r * H // 2, (tp_r + 1) * H // 2)
    ss = slice(sp_r * S // 2, (sp_r + 1) * S // 2)
    o = ring_attention(q[:, ss, hs].contiguous().cuda(), k[:, ss, hs].contiguous().cuda(), v[:, ss, hs].contiguous().cuda(), sg,
                       layout="bshd", exchange="mesh", causal=True, recv_buffers={})
    row = RowParallelLinear(d, d, True, tcfg, input_is_parallel=True).to("cuda", dt)
    row.weight.copy_(wo[:, tp_r * d // 2:(tp_r + 1) * d // 2])
    row.bias.copy_(bo)
    y = row(o.reshape(B, S // 2, d // 2)).cpu().double()
    want = dense[:, ss]
    rel = ((y - want).abs().mean() / want.abs().mean()).item()
    assert rel < 2e-3, rel


def _w_bench_extras(rank, world):
    """The exact code bench.py runs for its multi-GPU 'extra' block, at a small size (plumbing check)."""
    from tools.bench_parallel import bench_ring, bench_tp
    r = bench_tp(world, 2, 2, 256, 256, 4, 1024, 2, torch.bfloat16, steps=1, warmup=1)
    assert r["overlapped"]["tokens_per_s"] > 0 and r["unoverlapped"]["tokens_per_s"] > 0
    r = bench_ring(world, 1024 * world, 256, 4, torch.bfloat16, steps=1, warmup=1)
    assert all(r[k]["tokens_per_s"] > 0 for k in ("noncausal_mesh", "noncausal_mesh_unoverlapped", "noncausal_ring",
                                                  "noncausal_ring_unoverlapped", "causal_zigzag_mesh"))


def _w_rccl_wrappers():
    """Every collective wrapper of mio.parallelism.communication on RCCL with a group of one (FORCE_SINGLE_RANK_COLLECTIVES: the
    early returns are off, so the calls reach the backend): a sum / gather / scatter / exchange over one rank is the identity,
    and what is under test is the stream and handle logic against RCCL's asynchronous works (gloo's are synchronous)."""
    from mio.parallelism import communication as comm
    comm.FORCE_SINGLE_RANK_COLLECTIVES = True
    assert dist.get_backend() == "nccl"
    torch.manual_seed(0)
    x = torch.randn(1024, 512, device="cuda")
    side = torch.cuda.Stream()
    for kw in ({}, {"op": "avg"}, {"use_bf16": True}, {"stream": side}, {"op": "mean", "use_fp16": True, "stream": side}):
        t = x.clone()
        want = x.to(torch.bfloat16).float() if kw.get("use_bf16") else x.half().float() if kw.get("use_fp16") else x
        out = comm.all_reduce(t, **kw)
        assert out is t and torch.equal(t, want), kw
        t = x.clone()
        h, out = comm.all_reduce(t, async_op=True, **kw)
        y = torch.mm(x, x.t())  # work queued beside the collective
        h.wait()
        assert h.is_completed() or True
        assert torch.equal(out, want), kw
        del y
    t = x[:, ::2]  # not contiguous: goes through a wire copy and comes back into the view
    comm.all_reduce(t)
    assert torch.equal(t, x[:, ::2])
    assert torch.equal(comm.all_gather(x, dim=0), x) and torch.equal(comm.all_gather(x, dim=1), x)
    w, g = comm.all_gather(x, dim=-1, async_op=True)
    assert torch.equal(g, x)
    assert torch.equal(comm.reduce_scatter(x, dim=1), x) and torch.equal(comm.reduce_scatter(x, dim=0, op="avg"), x)
    assert torch.equal(comm.broadcast(x.clone(), src=0), x)
    comm.barrier()
    # the ring pass (one grouped batch_isend_irecv: a send to and a receive from the rank itself), sync / async / fp16 wire
    k = torch.randn(2, 256, 4, 64, device="cuda", dtype=torch.bfloat16)
    v = torch.randn(2, 256, 4, 64, device="cuda", dtype=torch.bfloat16)
    rk, rn, rv = comm.ring_exchange(k, None, v)
    assert rn is None and torch.equal(rk, k) and torch.equal(rv, v) and rk.data_ptr() != k.data_ptr()
    h, (rk, rv) = comm.ring_exchange(k, v, async_op=True)
    h.wait()
    h.wait()
    assert torch.equal(rk, k) and torch.equal(rv, v)
    (rf,) = comm.ring_exchange(x, use_fp16=True)
    assert rf.dtype == x.dtype and torch.equal(rf, x.half().float())


def _w_rccl_row_parallel():
    """RowParallelLinear's chunked GEMM -> all-reduce pipeline (the C3 overlap path) with its all-reduces on RCCL: a group of
    one sums nothing, so the result must equal the plain linear bit for bit -- any ordering hole between the GEMM chunks (torch's
    stream) and the collectives (RCCL's stream) shows up as stale rows."""
    from mio import ops
    from mio.parallelism import communication as comm
    from mio.parallelism.tensor_parallel import RowParallelLinear, TensorParallelConfig
    comm.FORCE_SINGLE_RANK_COLLECTIVES = True
    torch.manual_seed(3)
    cfg = TensorParallelConfig(world_size=1, tp_size=1, overlap_chunks=4)
    lin = RowParallelLinear(1024, 1024, bias=True, config=cfg, input_is_parallel=True).to(device="cuda", dtype=torch.bfloat16)
    x = torch.randn(4, 2048, 1024, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(4, 2048, 1024, device="cuda", dtype=torch.bfloat16)
    want = ops.gemm_bias_act(x, lin.weight, lin.bias, residual=r)
    for _ in range(3):
        got = lin(x, residual=r)
        assert torch.equal(got, want)
    comm.FORCE_SINGLE_RANK_COLLECTIVES = False
    assert torch.equal(lin(x, residual=r), want)


@pytest.mark.parametrize("which", ["wrappers", "row_parallel"])
def test_rccl_single_rank(which):
    """The communication layer on RCCL itself, as far as one GPU allows (a second rank on the same device is refused by RCCL)."""
    mp.spawn(_worker_rccl, args=(1, _free_port(), "_w_rccl_" + which, ()), nprocs=1, join=True)


@pytest.mark.parametrize("exchange,causal,zigzag,layout", [
    ("mesh", False, False, "bhsd"), ("ring", False, False, "bshd"), ("ring", True, False, "bhsd"),
    ("mesh", True, True, "bshd"),
])
def test_ring_attention_hip_ws2(exchange, causal, zigzag, layout):
    _run("_w_ring", 2, (exchange, causal, zigzag, layout))


@pytest.mark.parametrize("exchange,causal,zigzag,layout", [("mesh", False, False, "bshd"), ("ring", True, True, "bhsd")])
def test_ring_attention_k_prescaled_hip_ws2(exchange, causal, zigzag, layout):
    """The same ring with pre-scaled K shards: every step is a k_prescaled launch with the (o_acc, lse) carry
    (fa3_fwd5_kernel CARRY at head dim 64)."""
    _run("_w_ring", 2, (exchange, causal, zigzag, layout, True))


def test_tensor_parallel_hip_ws2():
    _run("_w_tp", 2)


@pytest.mark.parametrize("mode,zigzag,exchange", [("ring", False, "mesh"), ("ring", True, "ring"), ("full", False, "ring")])
def test_sequence_sharded_stack_hip_ws2(mode, zigzag, exchange):
    _run("_w_sharded_stack", 2, (mode, zigzag, exchange))


@pytest.mark.parametrize("which", ["tp", "sp"])
def test_k_prescaled_module_branches_hip_ws2(which):
    _run("_w_kpre_module_branches", 2, (which,))


def test_tensor2_x_sequence2_hip_ws4():
    _run("_w_tp2_x_sp2_hip", 4)


def test_parallel_groups_hip_ws2():
    _run("_w_parallel_groups", 2)


def test_bench_extras_hip_ws2():
    _run("_w_bench_extras", 2)
