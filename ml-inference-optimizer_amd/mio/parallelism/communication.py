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


# Tests only (tests/test_gpu_parallel.py::test_rccl_single_rank_*): a one-GPU box can host ONE RCCL rank, so the only way to run these
# wrappers' stream / handle logic on the real backend there is to let a group of one go through the collective instead of
# returning early.  Never set by the package itself.
FORCE_SINGLE_RANK_COLLECTIVES = False


def _alone(group: Optional[dist.ProcessGroup] = None) -> bool:
    """True when there is nobody to talk to: no process group, or a group of one (the reference's early returns)."""
    if not dist.is_initialized():
        return True
    return get_world_size(group) == 1 and not FORCE_SINGLE_RANK_COLLECTIVES


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
    if _alone(group):
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

    side = stream if (stream is not None and tensor.is_cuda) else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            work = dist.all_reduce(comm, op=rop, group=group, async_op=async_op)
            if not async_op:
                _finish()
        if not async_op:
            torch.cuda.current_stream().wait_stream(side)
            return tensor
        return _AsyncReduce(work, _finish, side), tensor
    work = dist.all_reduce(comm, op=rop, group=group, async_op=async_op)
    if async_op:
        return _AsyncReduce(work, _finish, None), tensor
    _finish()
    return tensor


class _AsyncReduce:
    """Handle of an asynchronous all_reduce: wait() = the collective is ordered before what the caller launches next
    AND the post-processing (copy back from the wire dtype, the division of "avg") has run -- on the side stream when
    one was given.  (A bare `work` would hand the caller an un-averaged / un-copied tensor.)"""

    def __init__(self, work, finish, stream):
        self._work, self._finish, self._stream, self._done = work, finish, stream, False

    def wait(self):
        if self._done:
            return True
        if self._stream is not None:
            with torch.cuda.stream(self._stream):
                self._work.wait()
                self._finish()
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            self._work.wait()
            self._finish()
        self._done = True
        return True

    def is_completed(self):
        return self._done or self._work.is_completed()


def all_gather(tensor: torch.Tensor, dim: int = 0, async_op: bool = False,
               group: Optional[dist.ProcessGroup] = None):
    """Gather along `dim` (reference :211-246).  Gathers straight into the pre-laid-out result when
    dim == 0 (no torch.cat); other dims gather then move the axis once."""
    ws = get_world_size(group)
    if _alone(group):
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
    if _alone(group):
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
    if _alone(group):
        return (None, tensor) if async_op else tensor
    work = dist.broadcast(tensor, src=src, group=group, async_op=async_op)
    return (work, tensor) if async_op else tensor


def barrier(group: Optional[dist.ProcessGroup] = None) -> None:
    if not _alone(group):
        dist.barrier(group=group)


def _strided_groups(world_size: int, size: int, stride: int, tag: str) -> Optional[dist.ProcessGroup]:
    """Groups {base + i * stride, i < size} covering all ranks (stride 1 = the reference's contiguous rank blocks,
    :493-498, :607-619); every rank creates every group (new_group is collective) and keeps its own.  Cached so
    repeated module construction is free."""
    if not dist.is_initialized():
        raise RuntimeError("Distributed environment not initialized. Call initialize_distributed first.")
    if world_size % (size * stride) != 0:
        raise ValueError(f"group size ({size}) x stride ({stride}) must divide world size ({world_size})")
    key = (tag, world_size, size, stride)
    if key in _GROUP_CACHE:
        return _GROUP_CACHE[key]
    rank = dist.get_rank()
    mine = None
    for outer in range(world_size // (size * stride)):
        for inner in range(stride):
            ranks = [outer * size * stride + inner + i * stride for i in range(size)]
            grp = dist.new_group(ranks)
            if rank in ranks:
                mine = grp
    _GROUP_CACHE[key] = mine
    return mine


def _registered(kind: str):
    """(True, group) when parallel_utils.initialize_parallel_groups has registered the data x sequence x tensor mesh."""
    from . import parallel_utils
    if parallel_utils._PARALLEL_GROUPS:
        return True, parallel_utils._PARALLEL_GROUPS.get(kind)
    return False, None


def setup_device_groups(world_size: int, tp_size: int) -> Optional[dist.ProcessGroup]:
    """Tensor-parallel group of this rank (reference :464-500): the registered mesh group when
    initialize_parallel_groups was called, else adjacent ranks (the mesh's innermost dimension)."""
    reg, grp = _registered("tensor")
    if reg:
        if (1 if grp is None else dist.get_world_size(grp)) != tp_size:
            raise ValueError(f"tp_size {tp_size} does not match the registered tensor-parallel group")
        return grp
    if tp_size > 1 and any(k[0] == "sp" and k[2] > 1 and k[3] != tp_size for k in _GROUP_CACHE):
        raise RuntimeError("tensor-parallel groups of adjacent ranks would share ranks with the sequence-parallel groups "
                           "already built: call parallel_utils.initialize_parallel_groups(ParallelConfig(...)) first, or "
                           "give SequenceParallelConfig the tp_size")
    return _strided_groups(world_size, tp_size, 1, "tp")


def setup_sequence_parallel_group(world_size: int, sp_size: int, tp_size: int = 1) -> Optional[dist.ProcessGroup]:
    """Sequence-parallel group of this rank (reference :580-619).  With tensor parallelism in the same job the
    sequence dimension strides by tp_size (mesh rank = (dp * SP + sp) * TP + tp): the reference builds both kinds from
    contiguous blocks, which makes a tp x sp job exchange K/V between ranks that hold different heads of the SAME
    tokens.  Resolution order: the mesh registered by initialize_parallel_groups; else stride tp_size."""
    reg, grp = _registered("sequence")
    if reg:
        if (1 if grp is None else dist.get_world_size(grp)) != sp_size:
            raise ValueError(f"sp_size {sp_size} does not match the registered sequence-parallel group")
        return grp
    if sp_size > 1 and tp_size == 1 and any(k[0] == "tp" and k[2] > 1 for k in _GROUP_CACHE):
        raise RuntimeError("sequence-parallel groups of contiguous ranks would share ranks with the tensor-parallel groups "
                           "already built: call parallel_utils.initialize_parallel_groups(ParallelConfig(...)) first, or "
                           "give SequenceParallelConfig the tp_size")
    return _strided_groups(world_size, sp_size, tp_size, "sp")


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
    if _alone(group):
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
        _done = False

        def wait(self_inner):
            if self_inner._done:  # a work is waited for once: gloo blocks on a second wait
                return
            self_inner._done = True
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


def mesh_exchange_start(tensors: Sequence[torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                        buffers: Optional[dict] = None):
    """Full-mesh variant of the ring pass for an 8-GPU xGMI node: rank r sends its tensors to EVERY
    peer and receives every peer's tensors, all posted in one grouped call so each of the 7 links
    carries one transfer concurrently (a neighbour ring would use 1 of the 7 links, hop by hop).
    Returns (handle, chunks) where chunks[i] is the list of tensors that originated on rank (r - i) % ws
    (chunks[0] = the local tensors), i.e. the same order a ring pass would deliver them in.

    handle.wait_chunk(i) orders chunk i's arrival before what the caller launches next.  Backends that return one
    work per transfer (gloo) wait per origin; RCCL coalesces a grouped call into ONE work, so the first wait_chunk
    covers every peer -- deliberately: per-origin completion on RCCL would need one group call per peer, and those run
    one after the other on the communicator's stream, i.e. one link at a time (7 x the transfer time of the one-group
    form, which already lands every chunk in about the time the local chunk's attention takes).

    buffers: a dict the caller keeps between calls (SequenceParallelConfig.buffer_reuse): the ws - 1 receive buffers
    are allocated once per (shape, dtype, device) instead of once per call."""
    ws = get_world_size(group)
    local = [t.contiguous() for t in tensors]
    if not dist.is_initialized() or ws == 1:
        return None, [local]
    r = get_rank(group)
    key = tuple((tuple(t.shape), t.dtype, str(t.device)) for t in local) + (ws,)
    pool = None if buffers is None else buffers.get(key)
    if pool is None:
        pool = [[torch.empty_like(t) for t in local] for _ in range(ws - 1)]
        if buffers is not None:
            buffers.clear()  # one live shape per module: a new shape replaces the old pool
            buffers[key] = pool
    chunks: List[List[torch.Tensor]] = [local]
    ops_ = []
    for i in range(1, ws):
        src = (r - i) % ws
        dst = (r + i) % ws
        gsrc = dist.get_global_rank(group, src) if group is not None else src
        gdst = dist.get_global_rank(group, dst) if group is not None else dst
        bufs = pool[i - 1]
        chunks.append(bufs)
        for t, b in zip(local, bufs):
            ops_.append(dist.P2POp(dist.isend, t, gdst, group))
            ops_.append(dist.P2POp(dist.irecv, b, gsrc, group))
    works = dist.batch_isend_irecv(ops_)
    per_op = len(works) == len(ops_)
    n_t = 2 * len(local)  # works per origin

    class _Handle:
        def __init__(self_inner):
            self_inner._waited = set()

        def wait_chunk(self_inner, i: int):
            """Chunk i (origin (r - i) % ws) has arrived -- and, per-op backends, our send paired with it was posted."""
            if i <= 0 or i in self_inner._waited:
                return
            if per_op:
                for w in works[(i - 1) * n_t:i * n_t]:
                    w.wait()
                self_inner._waited.add(i)
            else:
                self_inner.wait()

        def wait(self_inner):
            """Everything posted has completed (a work is waited for once: gloo blocks on a second wait)."""
            if per_op:
                for i in range(1, ws):
                    self_inner.wait_chunk(i)
            elif not self_inner._waited:
                for w in works:
                    w.wait()
                self_inner._waited.update(range(1, ws))

    return _Handle(), chunks
