"""Shape/partition helpers of the parallel wrappers (reference parallelism/parallel_utils.py:
ensure_divisibility/divide :11-40, split_tensor_along_dim :137-174, gather_tensor_along_dim :176-215,
set_tensor_model_parallel_attributes :491-514, get_partition_start_end :386-412)."""
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
