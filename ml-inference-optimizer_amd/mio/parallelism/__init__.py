from .communication import (initialize_distributed, get_rank, get_world_size, all_reduce, all_gather,  # noqa: F401
                            reduce_scatter, broadcast, barrier, ring_exchange, mesh_exchange_start,
                            scatter_along_sequence_dim, gather_along_sequence_dim, setup_device_groups,
                            setup_sequence_parallel_group)
from .tensor_parallel import (TensorParallelConfig, ColumnParallelLinear, RowParallelLinear,  # noqa: F401
                              TensorParallelMLP, TensorParallelAttention, ModelParallelConverter)
from .sequence_parallel import (SequenceParallelConfig, SequenceParallelAttention, SequenceParallelMLP,  # noqa: F401
                                SequenceShardedModule, SequenceParallelConverter, ring_attention,
                                partition_sequence, gather_sequence, zigzag_shard, zigzag_unshard)
