"""world_size-2 (and 4) gloo tests of the distributed schedules on CPU: tensor-parallel sharding +
all-reduce, ring / mesh K-V exchange with (o, lse) carry, zig-zag causal placement, sequence sharding.
The per-rank compute is the CPU checker (tests/_cpu_local.py); what is under test is the schedule.
Criterion = the reference's stated one (test_parallelism.py:306-322: sharded == unsharded), at 1e-5."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, fn_name, args):
    for p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    torch.set_grad_enabled(False)  # inference path: the HIP kernels carry no autograd
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import _cpu_local
        _cpu_local.install()
        globals()[fn_name](rank, world, *args)
    finally:
        dist.destroy_process_group()


def _run(fn_name, world=2, args=()):
    mp.spawn(_worker, args=(world, _free_port(), fn_name, args), nprocs=world, join=True)


# ---------------------------------------------------------------- workers (run inside the spawned ranks)
def _w_collectives(rank, world):
    from mio.parallelism import communication as comm
    t = torch.full((4, 3), float(rank + 1))
    assert torch.equal(comm.all_reduce(t.clone(), "sum"), torch.full((4, 3), float(sum(range(1, world + 1)))))
    assert torch.allclose(comm.all_reduce(t.clone(), "avg"), torch.full((4, 3), (world + 1) / 2.0))
    # asynchronous form with a wire dtype and "avg": the handle's wait() must copy back and divide
    t32 = torch.full((4, 3), float(rank + 1))
    h, same = comm.all_reduce(t32, "avg", async_op=True, use_bf16=True)
    assert same is t32
    h.wait()
    assert torch.allclose(t32, torch.full((4, 3), (world + 1) / 2.0)) and h.is_completed()
    h, t2 = comm.all_reduce(torch.full((2,), float(rank)), "sum", async_op=True)
    h.wait()
    assert t2[0].item() == sum(range(world))
    g = comm.all_gather(torch.full((2, 3), float(rank)), dim=1)
    assert g.shape == (2, 3 * world) and all(torch.equal(g[:, 3 * r:3 * r + 3], torch.full((2, 3), float(r))) for r in range(world))
    rs = comm.reduce_scatter(torch.arange(4 * world, dtype=torch.float32).view(2 * world, 2), dim=0)
    exp = torch.arange(4 * world, dtype=torch.float32).view(2 * world, 2)[2 * rank:2 * rank + 2] * world
    assert torch.equal(rs, exp)
    (got,) = comm.ring_exchange(torch.full((3,), float(rank)))
    assert torch.equal(got, torch.full((3,), float((rank - 1) % world)))
    h, chunks = comm.mesh_exchange_start([torch.full((2,), float(rank))])
    h.wait()
    for i in range(world):
        assert torch.equal(chunks[i][0], torch.full((2,), float((rank - i) % world)))
    # per-origin waits + receive buffers kept between calls (SequenceParallelConfig.buffer_reuse)
    pool = {}
    h, chunks = comm.mesh_exchange_start([torch.full((2,), float(rank)), torch.full((3,), 10.0 + rank)], buffers=pool)
    for i in range(1, world):
        h.wait_chunk(i)
        assert torch.equal(chunks[i][0], torch.full((2,), float((rank - i) % world)))
        assert torch.equal(chunks[i][1], torch.full((3,), 10.0 + (rank - i) % world))
    h.wait()
    first = [b.data_ptr() for c in chunks[1:] for b in c]
    dist.barrier()
    h, chunks = comm.mesh_exchange_start([torch.full((2,), 5.0 * rank), torch.full((3,), 7.0 + rank)], buffers=pool)
    h.wait()
    assert [b.data_ptr() for c in chunks[1:] for b in c] == first and len(pool) == 1
    assert torch.equal(chunks[world - 1][0], torch.full((2,), 5.0 * ((rank + 1) % world)))
    x = torch.arange(2 * 8 * 3, dtype=torch.float32).view(2, 8, 3)
    loc = comm.scatter_along_sequence_dim(x, world)
    assert torch.equal(loc, x[:, rank * (8 // world):(rank + 1) * (8 // world)])
    assert torch.equal(comm.gather_along_sequence_dim(loc, world), x)


def _w_tensor_parallel(rank, world):
    import oracle
    from mio.parallelism import TensorParallelConfig, TensorParallelMLP, TensorParallelAttention, ColumnParallelLinear, RowParallelLinear
    torch.manual_seed(0)  # same full weights on every rank
    d, I, H, B, S = 32, 64, 4, 2, 24
    cfg = TensorParallelConfig(world_size=world, tp_size=world, overlap_chunks=3)
    x = torch.randn(B, S, d)
    w1, b1, w2, b2 = torch.randn(I, d) * .2, torch.randn(I) * .2, torch.randn(d, I) * .2, torch.randn(d) * .2
    mlp = TensorParallelMLP(d, I, cfg, activation="gelu")
    per = I // world
    with torch.no_grad():
        mlp.dense_h_to_4h.weight.copy_(w1[rank * per:(rank + 1) * per]); mlp.dense_h_to_4h.bias.copy_(b1[rank * per:(rank + 1) * per])
        mlp.dense_4h_to_h.weight.copy_(w2[:, rank * per:(rank + 1) * per]); mlp.dense_4h_to_h.bias.copy_(b2)
    res = torch.randn(B, S, d)
    y = mlp(x, residual=res)
    ref = oracle.fused_mlp(x, w1, b1, w2, b2, "gelu", residual=res)
    assert (y.double() - ref).abs().max() < 1e-5
    assert torch.allclose(mlp.dense_h_to_4h.get_master_weight(), w1) and torch.allclose(mlp.dense_4h_to_h.get_master_weight(), w2)
    # attention: heads sharded, out-proj row-parallel
    att = TensorParallelAttention(d, H, cfg, causal=True)
    wq, wk, wv, wo = (torch.randn(d, d) * .2 for _ in range(4))
    bq, bk, bv, bo = (torch.randn(d) * .2 for _ in range(4))
    pd = d // world
    with torch.no_grad():
        for lin, w, b in ((att.query, wq, bq), (att.key, wk, bk), (att.value, wv, bv)):
            lin.weight.copy_(w[rank * pd:(rank + 1) * pd]); lin.bias.copy_(b[rank * pd:(rank + 1) * pd])
        att.output.weight.copy_(wo[:, rank * pd:(rank + 1) * pd]); att.output.bias.copy_(bo)
    y = att(x)
    F = torch.nn.functional
    q, k, v = (F.linear(x, w, b).view(B, S, H, d // H) for w, b in ((wq, bq), (wk, bk), (wv, bv)))
    ref = F.linear(oracle.standard_attention(q, k, v, causal=True).reshape(B, S, d).float(), wo, bo)
    assert (y - ref).abs().max() < 1e-4
    # column (gather_output) and row (input not parallel) on their own
    col = ColumnParallelLinear(d, I, config=cfg, gather_output=True)
    with torch.no_grad():
        col.weight.copy_(w1[rank * per:(rank + 1) * per]); col.bias.copy_(b1[rank * per:(rank + 1) * per])
    assert (col(x) - F.linear(x, w1, b1)).abs().max() < 1e-5
    row = RowParallelLinear(I, d, config=cfg, input_is_parallel=False)
    with torch.no_grad():
        row.weight.copy_(w2[:, rank * per:(rank + 1) * per]); row.bias.copy_(b2)
    h = torch.randn(B, S, I)
    dist.broadcast(h, 0)
    assert (row(h) - F.linear(h, w2, b2)).abs().max() < 1e-4


def _w_ring(rank, world, exchange, causal, zigzag, layout):
    import oracle
    from mio.parallelism.sequence_parallel import ring_attention, zigzag_shard
    torch.manual_seed(1)
    B, H, S, D = 2, 2, 16 * world, 8
    q, k, v = (torch.randn(B, S, H, D) for _ in range(3))  # full tensors, identical on every rank
    ref = oracle.standard_attention(q, k, v, causal=causal)
    if zigzag:
        loc = [zigzag_shard(t, rank, world, 1) for t in (q, k, v)]
        ref_loc = zigzag_shard(ref, rank, world, 1)
    else:
        n = S // world
        loc = [t[:, rank * n:(rank + 1) * n] for t in (q, k, v)]
        ref_loc = ref[:, rank * n:(rank + 1) * n]
    if layout == "bhsd":
        loc = [t.permute(0, 2, 1, 3).contiguous() for t in loc]
    else:
        loc = [t.contiguous() for t in loc]
    out = ring_attention(*loc, None, layout=layout, causal=causal, zigzag=zigzag, exchange=exchange)
    if layout == "bhsd":
        out = out.permute(0, 2, 1, 3)
    assert (out.double() - ref_loc).abs().max() < 1e-5, (exchange, causal, zigzag, layout)


def _w_ring_coalesced_work(rank, world):
    """RCCL returns ONE work for a grouped batch_isend_irecv; gloo one per transfer.  Force the coalesced shape under gloo
    (a wrapper whose single work waits for all of them) so that mesh_exchange_start's `per_op = False` branch -- the first
    wait_chunk covers every peer -- runs the whole ring_attention schedule (mesh, causal zig-zag and plain)."""
    import torch.distributed as d
    real = d.batch_isend_irecv

    class _All:
        def __init__(self, ws):
            self.ws = ws

        def wait(self):
            for w in self.ws:
                w.wait()
            self.ws = []

    d.batch_isend_irecv = lambda ops: [_All(real(ops))]
    try:
        _w_ring(rank, world, "mesh", True, True, "bshd")
        _w_ring(rank, world, "mesh", False, False, "bhsd")
    finally:
        d.batch_isend_irecv = real


def _w_ring_mask(rank, world):
    import oracle
    from mio.parallelism.sequence_parallel import ring_attention
    torch.manual_seed(2)
    B, H, S, D = 1, 2, 12 * world, 8
    q, k, v = (torch.randn(B, S, H, D) for _ in range(3))
    add = torch.randn(B, 1, S, S)
    ref = oracle.standard_attention(q, k, v, additive_mask=add)
    n = S // world
    sl = slice(rank * n, (rank + 1) * n)
    out = ring_attention(q[:, sl].contiguous(), k[:, sl].contiguous(), v[:, sl].contiguous(), None, layout="bshd",
                         exchange="ring", additive_mask=add[:, :, sl])
    assert (out.double() - ref[:, sl]).abs().max() < 1e-5


def _w_sp_modules(rank, world):
    import oracle
    from mio.parallelism import SequenceParallelConfig, SequenceParallelAttention, SequenceShardedModule
    torch.manual_seed(3)
    d, H, B, S = 32, 4, 2, 8 * world
    x = torch.randn(B, S, d)
    outs = {}
    for mode, causal, zz in (("ring", False, False), ("full", False, False), ("ring", True, True), ("full", True, True),
                             ("ring", True, False), ("full", True, False)):
        torch.manual_seed(4)
        cfg = SequenceParallelConfig(world_size=world, sp_size=world, attention_handling=mode, exchange="ring",
                                     causal=causal, zigzag=zz)
        att = SequenceParallelAttention(d, H, cfg, attention_dropout=0.0)
        y = SequenceShardedModule(att, cfg)(x)
        F = torch.nn.functional
        q, k, v = (F.linear(x, l.weight, l.bias).view(B, S, H, d // H) for l in (att.query, att.key, att.value))
        ref = F.linear(oracle.standard_attention(q, k, v, causal=causal).reshape(B, S, d).float(), att.output.weight,
                       att.output.bias)
        assert y.shape == x.shape and (y - ref).abs().max() < 1e-4, (mode, causal, zz)


def _w_sp_ring_hidden_1600(rank, world):
    """H * D = 1600 (GPT-2-XL: 25 heads of 64) is not a multiple of 128: the ring branch must not ask the K projection for
    a column scale (the checker's linear refuses such a range exactly as ops.gemm_bias_act does), ADVICE r2."""
    import oracle
    from mio.parallelism import SequenceParallelConfig, SequenceParallelAttention, SequenceShardedModule
    torch.manual_seed(5)
    d, H, B, S = 1600, 25, 1, 4 * world
    x = torch.randn(B, S, d) * 0.3
    cfg = SequenceParallelConfig(world_size=world, sp_size=world, attention_handling="ring", exchange="ring")
    att = SequenceParallelAttention(d, H, cfg, attention_dropout=0.0)
    y = SequenceShardedModule(att, cfg)(x)
    F = torch.nn.functional
    q, k, v = (F.linear(x, l.weight, l.bias).view(B, S, H, d // H) for l in (att.query, att.key, att.value))
    ref = F.linear(oracle.standard_attention(q, k, v).reshape(B, S, d).float(), att.output.weight, att.output.bias)
    assert (y - ref).abs().max() < 1e-4


# ---------------------------------------------------------------- tests
def _w_tp_x_sp(rank, world):
    """tensor (2) x sequence (2) mesh: heads split over the tensor group, the sequence over the sequence group; the
    attention block of every rank = its slice of the dense result, and the row-parallel out-projection summed over
    the tensor group = the dense block output for the rank's tokens."""
    from mio.parallelism import communication as comm
    from mio.parallelism.parallel_utils import (ParallelConfig, get_process_group_for_operation, group_ranks,
                                                initialize_parallel_groups)
    from mio.parallelism.sequence_parallel import ring_attention
    cfg = ParallelConfig(world, tensor_parallel_size=2, sequence_parallel_size=2)
    assert group_ranks(cfg, 3) == {"tensor": [2, 3], "sequence": [1, 3], "data": [3]}
    groups = initialize_parallel_groups(cfg)
    assert groups["data"] is None and get_process_group_for_operation("tensor") is groups["tensor"]
    tp_r, sp_r = comm.get_rank(groups["tensor"]), comm.get_rank(groups["sequence"])
    assert (tp_r, sp_r) == (rank % 2, rank // 2)
    one = torch.ones(1)
    assert comm.all_reduce(one.clone() * rank, "sum", group=groups["tensor"]).item() == sum(group_ranks(cfg, rank)["tensor"])
    assert comm.all_reduce(one.clone() * rank, "sum", group=groups["sequence"]).item() == sum(group_ranks(cfg, rank)["sequence"])

    torch.manual_seed(0)  # same tensors on every rank
    B, S, H, D = 1, 32, 4, 8
    d = H * D
    q, k, v = (torch.randn(B, H, S, D) for _ in range(3))
    wo = torch.randn(d, d) * 0.1
    dense = torch.softmax((q @ k.transpose(-1, -2)) / D ** 0.5, dim=-1) @ v   # [B,H,S,D]
    y_dense = dense.permute(0, 2, 1, 3).reshape(B, S, d) @ wo.T       # [B,S,d]
    hs = slice(tp_r * H // 2, (tp_r + 1) * H // 2)
    ss = slice(sp_r * S // 2, (sp_r + 1) * S // 2)
    o = ring_attention(q[:, hs, ss].contiguous(), k[:, hs, ss].contiguous(), v[:, hs, ss].contiguous(),
                       groups["sequence"], layout="bhsd", exchange="ring")
    assert torch.allclose(o, dense[:, hs, ss], atol=1e-5)
    part = o.permute(0, 2, 1, 3).reshape(B, S // 2, d // 2) @ wo[:, tp_r * d // 2:(tp_r + 1) * d // 2].T
    y = comm.all_reduce(part, "sum", group=groups["tensor"])
    assert torch.allclose(y, y_dense[:, ss], atol=1e-4)


def _w_tp_x_sp_configs(rank, world):
    """The config objects alone (no initialize_parallel_groups) on a tensor (2) x sequence (2) job: the tensor groups
    are adjacent ranks, the sequence groups stride by tp_size; a sequence config that does not know about the tensor
    groups is refused instead of silently ringing over ranks that hold different heads of the same tokens.  Then the
    block: ring attention over the sequence group on this rank's heads, row-parallel out-projection over the tensor
    group == the dense result for the rank's tokens."""
    import oracle
    from mio.parallelism import (RowParallelLinear, SequenceParallelConfig, TensorParallelConfig)
    from mio.parallelism import communication as comm
    from mio.parallelism.sequence_parallel import ring_attention
    tcfg = TensorParallelConfig(world_size=world, tp_size=2)
    tg = tcfg.get_tp_group()
    assert [dist.get_global_rank(tg, i) for i in range(2)] == [rank - rank % 2, rank - rank % 2 + 1]
    with pytest.raises(RuntimeError):
        SequenceParallelConfig(world_size=world, sp_size=2).get_sp_group()
    scfg = SequenceParallelConfig(world_size=world, sp_size=2, tp_size=2, exchange="mesh")
    sg = scfg.get_sp_group()
    assert sorted(dist.get_global_rank(sg, i) for i in range(2)) == [rank % 2, rank % 2 + 2]
    assert scfg.get_rank_info() == (rank // 2, 0) and tcfg.tp_rank() == rank % 2 and scfg.get_dp_size() == 1
    tp_r, sp_r = tcfg.tp_rank(), scfg.get_rank_info()[0]
    torch.manual_seed(0)
    B, S, H, D = 1, 32, 4, 8
    d = H * D
    q, k, v = (torch.randn(B, S, H, D) for _ in range(3))
    wo, bo = torch.randn(d, d) * 0.1, torch.randn(d) * 0.1
    dense = torch.nn.functional.linear(oracle.standard_attention(q, k, v).reshape(B, S, d).float(), wo, bo)
    hs = slice(tp_r * H // 2, (tp_r + 1) * H // 2)
    ss = slice(sp_r * S // 2, (sp_r + 1) * S // 2)
    o = ring_attention(q[:, ss, hs].contiguous(), k[:, ss, hs].contiguous(), v[:, ss, hs].contiguous(), sg,
                       layout="bshd", exchange="mesh", recv_buffers={})
    row = RowParallelLinear(d, d, True, tcfg, input_is_parallel=True)
    with torch.no_grad():
        row.weight.copy_(wo[:, tp_r * d // 2:(tp_r + 1) * d // 2])
        row.bias.copy_(bo)
    y = row(o.reshape(B, S // 2, d // 2))
    assert torch.allclose(y, dense[:, ss], atol=1e-4)


def test_collectives_ws2():
    _run("_w_collectives", 2)


def test_collectives_ws4():
    _run("_w_collectives", 4)


def test_tensor_parallel_ws2():
    _run("_w_tensor_parallel", 2)


@pytest.mark.parametrize("exchange,causal,zigzag,layout", [
    ("ring", False, False, "bhsd"), ("mesh", False, False, "bshd"), ("ring", True, False, "bshd"),
    ("mesh", True, True, "bhsd"),
])
def test_ring_attention_ws2(exchange, causal, zigzag, layout):
    _run("_w_ring", 2, (exchange, causal, zigzag, layout))


@pytest.mark.parametrize("exchange,causal,zigzag", [("ring", True, True), ("mesh", False, False), ("mesh", True, False)])
def test_ring_attention_ws4(exchange, causal, zigzag):
    _run("_w_ring", 4, (exchange, causal, zigzag, "bhsd"))


def test_ring_attention_mesh_coalesced_work_ws4():
    _run("_w_ring_coalesced_work", world=4)


def test_ring_attention_additive_mask_ws2():
    _run("_w_ring_mask", 2)


def test_sequence_parallel_modules_ws2():
    _run("_w_sp_modules", 2)


def test_sequence_parallel_ring_hidden_not_multiple_of_128_ws2():
    _run("_w_sp_ring_hidden_1600", 2)


def test_tensor_x_sequence_groups_ws4():
    _run("_w_tp_x_sp", world=4)


def test_tensor_x_sequence_configs_ws4():
    _run("_w_tp_x_sp_configs", world=4)
