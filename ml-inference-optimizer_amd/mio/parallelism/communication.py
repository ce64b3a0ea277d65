"""torch.distributed wrappers for the hot path (backend "nccl" == RCCL on ROCm; "gloo" on CPU).

Mirrors the call surface of reference parallelism/communication.py that the path uses:
initialize_distributed (:12-27), get_rank/get_world_size (:29-35), all_reduce (:37-209), all_gather
(:211-246), reduce_scatter (:248-304), broadcast (:306-332), barrier (:366-374), setup_device_groups
(:464-500), setup_sequence_parallel_group (:580-619), scatter/gather_along_sequence_dim (:621-698),
ring_exchange (second definition, :1694-1831).

MI355X notes: one process per GPU; the 8 GPUs of a node are a full xGMI mesh (7 links per GPU), so
besides the neighbour ring (`ring_exchange`) there is `mesh_exchange_start`, which posts every peer
transfer of a ring-attention pass at once (each pair uses its own link) instead of forwarding K/V hop
by hop.  Not carried over: the NVLink/NVLS NCCL env tuning (:886-1114) and the hand-rolled tree
all-reduce (:97-179) -- RCCL picks its own algorithm.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist

_OPS = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN,
        "prod": dist.ReduceOp.PRODUCT, "product": dist.ReduceOp.PRODUCT}
_GROUP_CACHE: Dict[Tuple, dist.ProcessGroup] = {}


def initialize_distributed(local_rank: Optional[int] = None, world_size: Optional[int] = None,
                           backend: str = "nccl") -> None:
    """Reference :12-27.  Arguments default to the torchrun environment (RANK/LOCAL_RANK/WORLD_SIZE)."""
    if dist.is_initialized():
        return
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size is None:
        world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", str(local_rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, world_size=world_size, rank=rank,
                                device_id=torch.device("cuda", local_rank))
    else:
        dist.init_process_group(backend=backend, world_size=world_size, rank=rank)


def get_rank(group: Optional[dist.ProcessGroup] = None) -> int:
    return dist.get_rank(group) if dist.is_initialized() else 0


def get_world_size(group: Optional[dist.ProcessGroup] = None) -> int:
    return dist.get_world_size(group) if dist.is_initialized() else 1


def _op(op) -> Tuple[dist.ReduceOp, bool]:
    if isinstance(op, str):
        name = op.lower()
        if name in ("avg", "mean"):
            return dist.ReduceOp.SUM, True
        if name not in _OPS:
            raise ValueError(f"Unsupported reduction operation: {op}")
        return _OPS[name], False
    return op, False


def all_reduce(tensor: torch.Tensor, op: Union[dist.ReduceOp, str] = dist.ReduceOp.SUM, async_op: bool = False,
               group: Optional[dist.ProcessGroup] = None, use_fp16: bool = False, use_bf16: bool = False,
               use_unbalanced: bool = False, stream: Optional[torch.cuda.Stream] = None):
    """In-place all-reduce (reference :37-209).  use_fp16/use_bf16 down-cast for the wire (:70-74);
    "avg" divides by the group size; use_unbalanced is accepted and ignored (RCCL picks tree/ring)."""
    if not dist.is_initialized() or get_world_size(group) == 1:
        return (None, tensor) if async_op else tensor
    rop, avg = _op(op)
    comm = tensor
    if tensor.is_floating_point():
        if use_fp16 and tensor.dtype != torch.float16:
            comm = tensor.to(torch.float16)
        elif use_bf16 and tensor.dtype != torch.bfloat16:
            comm = tensor.to(torch.bfloat16)
    if not comm.is_contiguous():
        comm = comm.contiguous()

    def _finish():
        if comm is not tensor:
            tensor.copy_(comm.to(tensor.dtype).view_as(tensor))
        if avg:
            tensor.div_(get_world_size(group))

    if stream is not None and tensor.is_cuda:
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            work = dist.all_reduce(comm, op=rop, group=group, async_op=async_op)
            if not async_op:
                _finish()
        if not async_op:
            torch.cuda.current_stream().wait_stream(stream)
            return tensor
        return work, tensor
    work = dist.all_reduce(comm, op=rop, group=group, async_op=async_op)
    if async_op:
        if comm is not tensor or avg:
            work.wait()
            _finish()
        return work, tensor
    _finish()
    return tensor


def all_gather(tensor: torch.Tensor, dim: int = 0, async_op: bool = False,
               group: Optional[dist.ProcessGroup] = None):
    """Gather along `dim` (reference :211-246).  Gathers straight into the pre-laid-out result when
    dim == 0 (no torch.cat); other dims gather then move the axis once."""
    ws = get_world_size(group)
    if not dist.is_initialized() or ws == 1:
        return (None, tensor) if async_op else tensor
    t = tensor.contiguous()
    dim = dim % t.dim()
    if t.dim() == 0:
        t = t.view(1)
    flat = torch.empty((ws * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    work = dist.all_gather_into_tensor(flat, t, group=group, async_op=async_op)
    if async_op:
        work.wait()
    out = flat.view((ws,) + tuple(t.shape))
    if dim == 0:
        res = out.view(ws * t.shape[0], *t.shape[1:])
    else:
        res = out.movedim(0, dim).reshape(*t.shape[:dim], ws * t.shape[dim], *t.shape[dim + 1:])
    return (work, res) if async_op else res


def reduce_scatter(tensor: torch.Tensor, dim: int = 0, op: Union[dist.ReduceOp, str] = dist.ReduceOp.SUM,
                   async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
    """Reduce then keep this rank's 1/ws slice along `dim` (reference :248-304)."""
    ws = get_world_size(group)
    if not dist.is_initialized() or ws == 1:
        return (None, tensor) if async_op else tensor
    rop, avg = _op(op)
    dim = dim % tensor.dim()
    if tensor.shape[dim] % ws != 0:
        raise ValueError(f"dimension {dim} of size {tensor.shape[dim]} is not divisible by world size {ws}")
    src = tensor.movedim(dim, 0).contiguous()
    out = torch.empty((src.shape[0] // ws,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if dist.get_backend(group) == "gloo":  # gloo has no reduce_scatter: all_reduce + slice (tests only)
        dist.all_reduce(src, op=rop, group=group)
        r = get_rank(group)
        out.copy_(src[r * out.shape[0]:(r + 1) * out.shape[0]])
        work = None
    else:
        work = dist.reduce_scatter_tensor(out, src, op=rop, group=group, async_op=async_op)
        if async_op:
            work.wait()
    if avg:
        out.div_(ws)
    res = out.movedim(0, dim)
    return (work, res) if async_op else res


def broadcast(tensor: torch.Tensor, src: int = 0, async_op: bool = False, group: Optional[dist.ProcessGroup] = None):
    if not dist.is_initialized() or get_world_size(group) == 1:
        return (None, tensor) if async_op else tensor
    work = dist.broadcast(tensor, src=src, group=group, async_op=async_op)
    return (work, tensor) if async_op else tensor


def barrier(group: Optional[dist.ProcessGroup] = None) -> None:
    if dist.is_initialized() and get_world_size(group) > 1:
        dist.barrier(group=group)


def _contiguous_groups(world_size: int, size: int, tag: str) -> Optional[dist.ProcessGroup]:
    """Contiguous rank blocks of `size` (reference :493-498, :607-619); every rank creates every group
    (new_group is collective) and keeps its own.  Cached so repeated module construction is free."""
    if not dist.is_initialized():
        raise RuntimeError("Distributed environment not initialized. Call initialize_distributed first.")
    if world_size % size != 0:
        raise ValueError(f"group size ({size}) must divide world size ({world_size})")
    key = (tag, world_size, size)
    if key in _GROUP_CACHE:
        return _GROUP_CACHE[key]
    rank = dist.get_rank()
    mine = None
    for g in range(world_size // size):
        ranks = list(range(g * size, (g + 1) * size))
        grp = dist.new_group(ranks)
        if rank in ranks:
            mine = grp
    _GROUP_CACHE[key] = mine
    return mine


def setup_device_groups(world_size: int, tp_size: int) -> Optional[dist.ProcessGroup]:
    """Tensor-parallel group of this rank (reference :464-500)."""
    return _contiguous_groups(world_size, tp_size, "tp")


def setup_sequence_parallel_group(world_size: int, sp_size: int) -> Optional[dist.ProcessGroup]:
    """Sequence-parallel group of this rank (reference :580-619)."""
    return _contiguous_groups(world_size, sp_size, "sp")


def scatter_along_sequence_dim(tensor: torch.Tensor, sp_size: Optional[int] = None,
                               group: Optional[dist.ProcessGroup] = None, seq_dim: int = 1) -> torch.Tensor:
    """This rank's contiguous S/sp slice -- a local narrow, no communication (reference :621-661)."""
    sp = sp_size or get_world_size(group)
    if sp == 1:
        return tensor
    S = tensor.shape[seq_dim]
    if S % sp != 0:
        raise ValueError(f"Sequence length ({S}) must be divisible by sequence parallel size ({sp})")
    r = (get_rank(group) if group is not None else get_rank()) % sp
    return tensor.narrow(seq_dim, r * (S // sp), S // sp).contiguous()


def gather_along_sequence_dim(tensor: torch.Tensor, sp_size: Optional[int] = None,
                              group: Optional[dist.ProcessGroup] = None, seq_dim: int = 1) -> torch.Tensor:
    """All-gather the S/sp slices (reference :663-698; the reference gathers on WORLD, :690 -- here the
    sequence-parallel group is used when given)."""
    sp = sp_size or get_world_size(group)
    if sp == 1 or not dist.is_initialized():
        return tensor
    return all_gather(tensor, dim=seq_dim, group=group)


def ring_exchange(*tensors: torch.Tensor, group: Optional[dist.ProcessGroup] = None, async_op: bool = False,
                  use_fp16: bool = False, use_nccl_collectives: bool = True):
    """Each rank sends `tensors` to rank+1 and receives the same shapes from rank-1 (reference
    :1694-1831).  None entries pass through (the reference crashes on a None mask, :1738-1743).
    One grouped batch_isend_irecv -> a single RCCL group call; no clones of the send buffers."""
    if not dist.is_initialized() or get_world_size(group) == 1:
        return (None, list(tensors)) if async_op else list(tensors)
    ws, r = get_world_size(group), get_rank(group)
    nxt = dist.get_global_rank(group, (r + 1) % ws) if group is not None else (r + 1) % ws
    prv = dist.get_global_rank(group, (r - 1) % ws) if group is not None else (r - 1) % ws
    send, recv, ops_ = [], [], []
    for t in tensors:
        if t is None:
            send.append(None)
            recv.append(None)
            continue
        s = t.contiguous()
        if use_fp16 and s.is_floating_point() and s.dtype != torch.float16:
            s = s.to(torch.float16)
        send.append(s)
        recv.append(torch.empty_like(s))
    for s, rv in zip(send, recv):
        if s is None:
            continue
        ops_.append(dist.P2POp(dist.isend, s, nxt, group))
        ops_.append(dist.P2POp(dist.irecv, rv, prv, group))
    works = dist.batch_isend_irecv(ops_) if ops_ else []
    out = recv

    class _Handle:
        def wait(self_inner):
            for w in works:
                w.wait()
            if use_fp16:
                for i, t in enumerate(tensors):
                    if t is not None and out[i].dtype != t.dtype:
                        out[i] = out[i].to(t.dtype)

    h = _Handle()
    if async_op:
        h._keepalive = send
        return h, out
    h.wait()
    return out


def mesh_exchange_start(tensors: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None):
    """Full-mesh variant of the ring pass for an 8-GPU xGMI node: rank r sends its tensors to EVERY
    peer and receives every peer's tensors, all posted in one grouped call so each of the 7 links
    carries one transfer concurrently (a neighbour ring would use 1 of the 7 links, hop by hop).
    Returns (handle, chunks) where chunks[i] is the list of tensors that originated on rank (r - i) % ws
    (chunks[0] = the local tensors), i.e. the same order a ring pass would deliver them in."""
    ws = get_world_size(group)
    local = [t.contiguous() for t in tensors]
    if not dist.is_initialized() or ws == 1:
        return None, [local]
    r = get_rank(group)
    chunks: List[List[torch.Tensor]] = [local]
    ops_ = []
    for i in range(1, ws):
        src = (r - i) % ws
        dst = (r + i) % ws
        gsrc = dist.get_global_rank(group, src) if group is not None else src
        gdst = dist.get_global_rank(group, dst) if group is not None else dst
        bufs = [torch.empty_like(t) for t in local]
        chunks.append(bufs)
        for t, b in zip(local, bufs):
            ops_.append(dist.P2POp(dist.isend, t, gdst, group))
            ops_.append(dist.P2POp(dist.irecv, b, gsrc, group))
    works = dist.batch_isend_irecv(ops_)

    class _Handle:
        def wait(self_inner):
            for w in works:
                w.wait()

    return _Handle(), chunks
