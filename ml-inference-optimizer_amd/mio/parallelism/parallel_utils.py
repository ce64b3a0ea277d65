"""Shape/partition helpers of the parallel wrappers (reference parallelism/parallel_utils.py:
ensure_divisibility/divide :11-40, split_tensor_along_dim :137-174, gather_tensor_along_dim :176-215,
set_tensor_model_parallel_attributes :491-514, get_partition_start_end :386-412, initialize_parallel_groups /
get_process_group_for_operation :882-1019; ParallelConfig: orchestrator.py:20-110)."""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

from . import communication as comm


def ensure_divisibility(numerator: int, denominator: int) -> None:
    if numerator % denominator != 0:
        raise ValueError(f"{numerator} is not divisible by {denominator}")


def divide(numerator: int, denominator: int) -> int:
    ensure_divisibility(numerator, denominator)
    return numerator // denominator


def split_tensor_along_dim(tensor: torch.Tensor, dim: int, world_size: Optional[int] = None,
                           contiguous: bool = True, num_partitions: Optional[int] = None) -> List[torch.Tensor]:
    """`num_partitions` is accepted as an alias: the reference's own callers pass it
    (tensor_parallel.py:293,752,759,789) although its signature lacks it (a TypeError there)."""
    n = num_partitions or world_size or comm.get_world_size()
    size = divide(tensor.shape[dim], n)
    parts = torch.split(tensor, size, dim=dim)
    return [p.contiguous() for p in parts] if contiguous else list(parts)


def gather_tensor_along_dim(tensor: torch.Tensor, dim: int, world_size: Optional[int] = None,
                            dest_rank: Optional[int] = None, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    if (world_size or comm.get_world_size(group)) == 1 or not dist.is_initialized():
        return tensor
    return comm.all_gather(tensor, dim=dim, group=group)


def get_partition_start_end(total: int, rank: int, world_size: int) -> Tuple[int, int]:
    per = divide(total, world_size)
    return rank * per, (rank + 1) * per


def set_tensor_model_parallel_attributes(tensor: torch.Tensor, is_parallel: bool, dim: int, stride: int) -> None:
    setattr(tensor, "tensor_model_parallel", is_parallel)
    setattr(tensor, "partition_dim", dim)
    setattr(tensor, "partition_stride", stride)


# ---------------------------------------------------------------------------------------------------------------
# combined tensor x sequence x data groups (reference parallel_utils.py:882-1019, orchestrator.py:20-110)
# ---------------------------------------------------------------------------------------------------------------
class ParallelConfig:
    """world_size = data x sequence x tensor (pipeline parallelism is outside the hot path: must be 1)."""

    def __init__(self, world_size: int, tensor_parallel_size: int = 1, sequence_parallel_size: int = 1,
                 data_parallel_size: int = 1, pipeline_parallel_size: int = 1,
                 communication_dtype: torch.dtype = torch.float16, overlap_communication: bool = True,
                 optimize_memory: bool = True, activation_checkpointing: bool = False):
        self.world_size = world_size
        self.tensor_parallel_size = tensor_parallel_size
        self.sequence_parallel_size = sequence_parallel_size
        self.data_parallel_size = data_parallel_size
        self.pipeline_parallel_size = pipeline_parallel_size
        self.communication_dtype = communication_dtype
        self.overlap_communication = overlap_communication
        self.optimize_memory = optimize_memory
        self.activation_checkpointing = activation_checkpointing
        if pipeline_parallel_size != 1:
            raise ValueError("pipeline parallelism is not part of this path (pipeline_parallel_size must be 1)")
        prod = tensor_parallel_size * sequence_parallel_size * data_parallel_size
        if prod != world_size:
            raise ValueError(f"tensor ({tensor_parallel_size}) x sequence ({sequence_parallel_size}) x data "
                             f"({data_parallel_size}) = {prod} does not equal world_size ({world_size})")


_PARALLEL_GROUPS: dict = {}


def group_ranks(config: ParallelConfig, rank: int) -> dict:
    """Ranks of `rank`'s tensor / sequence / data group on the mesh rank = (dp * SP + sp) * TP + tp.

    Tensor-parallel ranks are adjacent (their all-reduces are the most frequent exchange: on one node they are
    direct xGMI neighbours), sequence groups stride by TP, data groups by SP * TP.  (The reference builds the tensor
    AND the sequence groups from contiguous blocks, :905-945, so with both > 1 they contain the same ranks and a
    ring over "sequence" would exchange K/V between ranks that hold different heads of the SAME tokens.)"""
    tp, sp, dp = config.tensor_parallel_size, config.sequence_parallel_size, config.data_parallel_size
    t, s_, d = rank % tp, (rank // tp) % sp, rank // (tp * sp)
    return {"tensor": [(d * sp + s_) * tp + i for i in range(tp)],
            "sequence": [(d * sp + i) * tp + t for i in range(sp)],
            "data": [(i * sp + s_) * tp + t for i in range(dp)]}


def initialize_parallel_groups(config: ParallelConfig) -> dict:
    """Create (collectively: every rank creates every group) and return this rank's {"tensor", "sequence", "data"}
    process groups; a dimension of size 1 maps to None (= no communication)."""
    if not dist.is_initialized():
        raise RuntimeError("Distributed backend must be initialized before creating process groups")
    if dist.get_world_size() != config.world_size:
        raise ValueError(f"config.world_size ({config.world_size}) != dist world size ({dist.get_world_size()})")
    rank = dist.get_rank()
    mine: dict = {"tensor": None, "sequence": None, "data": None}
    sizes = {"tensor": config.tensor_parallel_size, "sequence": config.sequence_parallel_size,
             "data": config.data_parallel_size}
    for kind in ("tensor", "sequence", "data"):
        if sizes[kind] == 1:
            continue
        seen = set()
        for r in range(config.world_size):
            ranks = tuple(group_ranks(config, r)[kind])
            if ranks in seen:
                continue
            seen.add(ranks)
            grp = dist.new_group(list(ranks))
            if rank in ranks:
                mine[kind] = grp
    _PARALLEL_GROUPS.clear()
    _PARALLEL_GROUPS.update(mine)
    return mine


def get_process_group_for_operation(op_type: str):
    """Group registered by initialize_parallel_groups for "tensor" / "sequence" / "data" (None: size 1).  The
    reference returns WORLD for everything (:1004-1019)."""
    if op_type not in ("tensor", "sequence", "data"):
        raise ValueError(f"unknown operation type: {op_type}")
    if not dist.is_initialized():
        return None
    return _PARALLEL_GROUPS.get(op_type)
