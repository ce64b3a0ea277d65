"""Host-side mirror of the parts of the reference's baseline/inference.py that sit either side of the hot path
(SURVEY.md 8 b "Operator registration (ii)", 8 f 1 and 8 f 3):

  * FusionPattern / FusionRegistry / fusion_registry (inference.py:26-261) with WORKING Linear+GELU+Linear and
    Linear+ReLU+Linear patterns (the reference's fusion functions call constructor keywords and a
    `copy_weights_from` that its FusedMLP classes do not have, so they raise before fusing);
  * BlockManager / SequenceMetadata / PagedKVCache (inference.py:1045-1303): the allocator behind the paged decode
    kernel -- same 5-D cache layout [num_blocks, L, block_size, H, Dh] and int32 block tables the kernels read;
  * InferenceRunner.warmup / run_inference timing semantics (inference.py:616-713) and create_inference_runner.

Nothing here computes on the CPU: the modules the registry builds and the caches the allocator hands out are the
HIP-backed ones (mio.kernels.*, mio.ops).  Quantisation, diffusion runners and HF loaders are out of scope (8 f).
"""
from __future__ import annotations

import logging
import math
import time
from abc import ABC, abstractmethod
from copy import deepcopy
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn as nn

from ..kernels.attention.flash_attention import FlashAttentionConfig, ModelConverter
from ..kernels.mlp.fused_mlp import FusedMLP, FusedMLPConfig, FusedMLPGeluTanh, FusedMLPReLU


# ---------------------------------------------------------------------------------------------------------------
# fusion registry (inference.py:26-261)
# ---------------------------------------------------------------------------------------------------------------
class FusionPattern:
    """A sequence of module types that can be replaced by one fused module (inference.py:26-73)."""

    def __init__(self, name: str, pattern: List[Union[type, Tuple[type, ...]]], fusion_fn: Callable,
                 description: Optional[str] = None, predicate: Optional[Callable[[List[nn.Module]], bool]] = None):
        self.name = name
        self.pattern = pattern
        self.fusion_fn = fusion_fn
        self.description = description or f"Fuses {[getattr(p, '__name__', str(p)) for p in pattern]}"
        self.predicate = predicate  # extra shape test (not in the reference): e.g. fc2.out == fc1.in

    def match(self, modules: List[nn.Module]) -> bool:
        if len(modules) != len(self.pattern):
            return False
        if not all(isinstance(m, p) for m, p in zip(modules, self.pattern)):
            return False
        return self.predicate is None or bool(self.predicate(modules))

    def fuse(self, modules: List[nn.Module]) -> nn.Module:
        return self.fusion_fn(modules)


class FusionRegistry:
    """Registry of fusion patterns (inference.py:76-215); same methods, same traversal order."""

    def __init__(self):
        self.patterns: List[FusionPattern] = []

    def register_pattern(self, pattern: FusionPattern) -> None:
        self.patterns.append(pattern)

    def find_matching_pattern(self, modules: List[nn.Module]) -> Optional[FusionPattern]:
        for pattern in self.patterns:
            if pattern.match(modules):
                return pattern
        return None

    def fuse_modules(self, model: nn.Module, inplace: bool = False) -> nn.Module:
        if not inplace:
            model = deepcopy(model)
        for _parent_name, parent, start, modules in reversed(self._find_fusion_candidates(model)):
            pattern = self.find_matching_pattern(modules)
            if pattern is None:
                continue
            child_names = list(dict(parent.named_children()).keys())[start:start + len(modules)]
            self._replace_modules(parent, child_names, pattern.fuse(modules))
        return model

    def _find_fusion_candidates(self, model: nn.Module) -> List[Tuple[str, nn.Module, int, List[nn.Module]]]:
        candidates = []
        if not self.patterns:
            return candidates
        max_len = max(len(p.pattern) for p in self.patterns)
        for parent_name, parent in model.named_modules():
            children = list(parent.children())
            if not children:
                continue
            i = 0
            while i < len(children):  # non-overlapping, left to right (the reference may return overlapping runs)
                hit = 0
                for n in range(min(max_len, len(children) - i), 1, -1):
                    if self.find_matching_pattern(children[i:i + n]):
                        candidates.append((parent_name, parent, i, children[i:i + n]))
                        hit = n
                        break
                i += hit if hit else 1
        return candidates

    def _replace_modules(self, parent: nn.Module, child_names: List[str], new_module: nn.Module) -> None:
        if isinstance(parent, nn.Sequential):
            old = list(parent.named_children())
            keep_before = [m for n, m in old if n not in child_names and
                           [k for k, _ in old].index(n) < [k for k, _ in old].index(child_names[0])]
            keep_after = [m for n, m in old if n not in child_names and
                          [k for k, _ in old].index(n) > [k for k, _ in old].index(child_names[-1])]
            for key in list(parent._modules.keys()):
                del parent._modules[key]
            for m in keep_before + [new_module] + keep_after:
                parent.add_module(str(len(parent)), m)
        else:  # attribute-style containers (transformer blocks): first name takes the fused module
            for name in child_names[1:]:
                if hasattr(parent, name):
                    delattr(parent, name)
            if hasattr(parent, child_names[0]):
                setattr(parent, child_names[0], new_module)


fusion_registry = FusionRegistry()


def _mlp_shapes_ok(modules: List[nn.Module]) -> bool:
    fc1, _act, fc2 = modules
    return fc1.out_features == fc2.in_features and fc2.out_features == fc1.in_features


def _fused_from_linears(cls, fc1: nn.Linear, fc2: nn.Linear, activation_fn: str) -> nn.Module:
    precision = {torch.float16: "fp16", torch.bfloat16: "bf16"}.get(fc1.weight.dtype, "bf16")
    mlp = cls(fc1.in_features, fc1.out_features, FusedMLPConfig(activation_fn=activation_fn, precision=precision))
    mlp = mlp.to(device=fc1.weight.device, dtype=fc1.weight.dtype)
    with torch.no_grad():
        mlp.fc1.weight.copy_(fc1.weight)
        mlp.fc2.weight.copy_(fc2.weight)
        for dst, src in ((mlp.fc1, fc1), (mlp.fc2, fc2)):
            if src.bias is not None:
                dst.bias.copy_(src.bias)
            else:
                dst.bias.zero_()
    return mlp


def fuse_mlp_gelu(modules: List[nn.Module]) -> nn.Module:
    """Linear + GELU + Linear (inference.py:228-243).  nn.GELU(approximate='tanh') -> tanh kernel; the default exact
    GELU -> erf kernel, so the fused module computes what the unfused one did (the reference always picks the tanh
    form, which changes the numerics of an exact-GELU model)."""
    fc1, gelu, fc2 = modules
    if getattr(gelu, "approximate", "none") == "tanh":
        return _fused_from_linears(FusedMLPGeluTanh, fc1, fc2, "gelu")
    return _fused_from_linears(FusedMLP, fc1, fc2, "gelu")


def fuse_mlp_relu(modules: List[nn.Module]) -> nn.Module:
    """Linear + ReLU + Linear (inference.py:246-261)."""
    fc1, _relu, fc2 = modules
    return _fused_from_linears(FusedMLPReLU, fc1, fc2, "relu")


fusion_registry.register_pattern(FusionPattern(
    name="linear_gelu_linear", pattern=[nn.Linear, nn.GELU, nn.Linear], fusion_fn=fuse_mlp_gelu,
    description="Fuses Linear + GELU + Linear into a single fused-MLP module", predicate=_mlp_shapes_ok))
fusion_registry.register_pattern(FusionPattern(
    name="linear_relu_linear", pattern=[nn.Linear, nn.ReLU, nn.Linear], fusion_fn=fuse_mlp_relu,
    description="Fuses Linear + ReLU + Linear into a single FusedMLPReLU module", predicate=_mlp_shapes_ok))


def convert_to_flash_attention(model: nn.Module) -> nn.Module:
    """inference.py:283-304: replace attention modules by the flash-attention layers."""
    return ModelConverter(FlashAttentionConfig()).convert_model(model)


# ---------------------------------------------------------------------------------------------------------------
# paged KV cache allocator (inference.py:1045-1303)
# ---------------------------------------------------------------------------------------------------------------
class BlockManager:
    """Physical block pool with reference counts (inference.py:1045-1126).  Cache layout
    [num_blocks, num_layers, block_size, num_heads, head_dim] -- what mio_fa3_decode_paged / mio_reshape_and_cache
    read and write.  Reference counts live on the host (the reference keeps them in a device tensor and pays a
    device->host sync per allocate/free)."""

    def __init__(self, num_blocks: int, block_size: int, num_layers: int, num_heads: int, head_dim: int,
                 dtype: torch.dtype, device: str):
        self.num_blocks, self.block_size, self.num_layers = num_blocks, block_size, num_layers
        self.num_heads, self.head_dim, self.dtype, self.device = num_heads, head_dim, dtype, device
        self.free_blocks = list(range(num_blocks))
        self.ref_counts = [0] * num_blocks
        shape = (num_blocks, num_layers, block_size, num_heads, head_dim)
        self.gpu_cache_k = torch.zeros(shape, dtype=dtype, device=device)
        self.gpu_cache_v = torch.zeros(shape, dtype=dtype, device=device)
        self.is_initialized = True

    def allocate_block(self) -> int:
        if not self.free_blocks:
            raise MemoryError("Out of memory: No free blocks available in KV cache.")
        idx = self.free_blocks.pop()
        self.ref_counts[idx] = 1
        return idx

    def free_block(self, block_idx: int) -> None:
        if self.ref_counts[block_idx] <= 0:
            logging.warning(f"Attempting to free block {block_idx} with ref count {self.ref_counts[block_idx]}.")
            return
        self.ref_counts[block_idx] -= 1
        if self.ref_counts[block_idx] == 0:
            self.free_blocks.append(block_idx)

    def increase_ref_count(self, block_idx: int) -> None:
        if self.ref_counts[block_idx] <= 0:
            raise ValueError(f"Cannot increase ref count for unallocated block {block_idx}.")
        self.ref_counts[block_idx] += 1

    def get_num_free_blocks(self) -> int:
        return len(self.free_blocks)

    def get_physical_block(self, block_idx: int, layer_idx: int) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.gpu_cache_k[block_idx, layer_idx], self.gpu_cache_v[block_idx, layer_idx]

    def get_physical_caches(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.gpu_cache_k, self.gpu_cache_v


class SequenceMetadata:
    """Block table of one sequence (inference.py:1129-1147)."""

    def __init__(self, seq_id: int):
        self.seq_id = seq_id
        self.logical_len = 0
        self.block_table: List[int] = []

    def append_block(self, block_idx: int):
        self.block_table.append(block_idx)

    def get_last_block_physical_idx(self) -> Optional[int]:
        return self.block_table[-1] if self.block_table else None

    def __len__(self) -> int:
        return len(self.block_table)


class PagedKVCache:
    """Logical block tables over a BlockManager (inference.py:1150-1303); same methods and error behaviour.
    `kernel_metadata` (not in the reference) packs the tables of a batch into the int32 tensors the kernels take."""

    def __init__(self, num_blocks: int, block_size: int, num_layers: int, num_heads: int, head_dim: int,
                 dtype: torch.dtype = torch.float16, device: str = "cuda"):
        self.block_manager = BlockManager(num_blocks, block_size, num_layers, num_heads, head_dim, dtype, device)
        self.block_size, self.num_layers, self.num_heads, self.head_dim = block_size, num_layers, num_heads, head_dim
        self.dtype, self.device = dtype, device
        self.sequences: Dict[int, SequenceMetadata] = {}
        self.prefix_cache: Dict[Tuple[int, ...], List[int]] = {}

    def _ensure_sequence_exists(self, seq_id: int):
        if seq_id not in self.sequences:
            self.sequences[seq_id] = SequenceMetadata(seq_id)

    def _get_logical_block_idx(self, token_pos: int) -> int:
        return token_pos // self.block_size

    def _get_block_offset(self, token_pos: int) -> int:
        return token_pos % self.block_size

    def allocate_blocks_for_sequence(self, seq_id: int, num_tokens: int):
        self._ensure_sequence_exists(seq_id)
        meta = self.sequences[seq_id]
        for _ in range(math.ceil(num_tokens / self.block_size) - len(meta)):
            try:
                meta.append_block(self.block_manager.allocate_block())
            except MemoryError:
                self.free_sequence(seq_id)
                raise
        meta.logical_len = num_tokens

    def append_token(self, seq_id: int) -> None:
        self._ensure_sequence_exists(seq_id)
        meta = self.sequences[seq_id]
        new_len = meta.logical_len + 1
        cur_blk = self._get_logical_block_idx(meta.logical_len - 1 if meta.logical_len > 0 else 0)
        if self._get_logical_block_idx(new_len - 1) > cur_blk or not meta.block_table:
            try:
                meta.append_block(self.block_manager.allocate_block())
            except MemoryError:
                self.free_sequence(seq_id)
                raise
        meta.logical_len = new_len

    def get_block_table(self, seq_id: int) -> List[int]:
        if seq_id not in self.sequences:
            raise ValueError(f"Sequence {seq_id} not found in cache.")
        return self.sequences[seq_id].block_table

    def get_sequence_length(self, seq_id: int) -> int:
        return self.sequences[seq_id].logical_len if seq_id in self.sequences else 0

    def free_sequence(self, seq_id: int) -> None:
        if seq_id in self.sequences:
            for blk in self.sequences[seq_id].block_table:
                self.block_manager.free_block(blk)
            del self.sequences[seq_id]
        else:
            logging.warning(f"Attempted to free non-existent sequence {seq_id}")

    def get_physical_caches(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.block_manager.get_physical_caches()

    def kernel_metadata(self, seq_ids: Sequence[int]) -> Tuple[torch.Tensor, torch.Tensor, int]:
        """(block_tables int32 [B, max_blocks], context_lengths int32 [B], max_seq_len) for a batch of sequences --
        the arguments of paged_attention_forward / reshape_and_cache (attention_kernels.py:1206-1216, 1313-1323)."""
        tables = [self.get_block_table(s) for s in seq_ids]
        width = max(1, max(len(t) for t in tables))
        bt = torch.zeros(len(tables), width, dtype=torch.int32)
        for i, t in enumerate(tables):
            bt[i, :len(t)] = torch.tensor(t, dtype=torch.int32)
        lens = [self.get_sequence_length(s) for s in seq_ids]
        cl = torch.tensor(lens, dtype=torch.int32)
        return bt.to(self.device), cl.to(self.device), max(lens) if lens else 0

    def get_memory_usage(self) -> Dict[str, float]:
        bm = self.block_manager
        total, free = bm.num_blocks, bm.get_num_free_blocks()
        phys = (bm.gpu_cache_k.element_size() * bm.gpu_cache_k.nelement() +
                bm.gpu_cache_v.element_size() * bm.gpu_cache_v.nelement()) / (1024 * 1024)
        return {"total_physical_blocks": total, "free_physical_blocks": free, "used_physical_blocks": total - free,
                "block_size": self.block_size, "total_physical_memory_mb": phys,
                "gpu_cache_k_shape": tuple(bm.gpu_cache_k.shape), "gpu_cache_v_shape": tuple(bm.gpu_cache_v.shape),
                "memory_efficiency": free / total if total > 0 else 1.0, "active_sequences": len(self.sequences)}


# ---------------------------------------------------------------------------------------------------------------
# inference runner (inference.py:377-713, 1779-1810): timing semantics only
# ---------------------------------------------------------------------------------------------------------------
class InferenceRunner(ABC):
    """warmup / run_inference with the reference's metric names (inference.py:616-713).  The model is moved to
    `device` and, for fp16 / bf16, cast to that dtype (:406-427); int8 / int4 quantisation is out of scope."""

    def __init__(self, model: nn.Module, device: str, precision: str = "fp16"):
        self.device, self.precision = device, precision
        dtype = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}.get(precision)
        if dtype is None:
            raise ValueError(f"Unsupported precision: {precision}")
        self.model = model.to(device=device, dtype=dtype)
        self.metrics: Dict[str, float] = {}

    def warmup(self, inputs: Any, iterations: int = 10) -> None:
        with torch.no_grad():
            self.model.eval()
            if self.device == "cuda":
                torch.cuda.reset_peak_memory_stats()
            for _ in range(iterations):
                self._forward(inputs)
            if self.device == "cuda":
                torch.cuda.synchronize()

    @abstractmethod
    def _forward(self, inputs: Any, **kwargs) -> Any:
        ...

    def run_inference(self, inputs: Any, **kwargs) -> Tuple[Any, Dict[str, float]]:
        self.model.eval()
        metrics: Dict[str, float] = {}
        cuda = self.device == "cuda"
        if cuda:
            torch.cuda.reset_peak_memory_stats()
            metrics["memory_before_mb"] = torch.cuda.memory_allocated() / (1024 ** 2)
            start_event, end_event = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            start_event.record()
        t0 = time.perf_counter()
        with torch.no_grad():
            outputs = self._forward(inputs, **kwargs)
        t1 = time.perf_counter()
        if cuda:
            end_event.record()
            torch.cuda.synchronize()
            metrics["cuda_time_ms"] = start_event.elapsed_time(end_event)
        metrics["total_time_ms"] = (t1 - t0) * 1000
        if cuda:
            metrics["memory_after_mb"] = torch.cuda.memory_allocated() / (1024 ** 2)
            metrics["peak_memory_mb"] = torch.cuda.max_memory_allocated() / (1024 ** 2)
            metrics["memory_change_mb"] = metrics["memory_after_mb"] - metrics["memory_before_mb"]
        self.metrics = metrics
        return outputs, metrics

    def run_batch_inference(self, batch_inputs: List[Any], **kwargs) -> List[Tuple[Any, Dict[str, float]]]:
        return [self.run_inference(x, **kwargs) for x in batch_inputs]


class BasicInferenceRunner(InferenceRunner):
    """model(inputs) / model(**inputs) (the reference's base runner `_forward`)."""

    def _forward(self, inputs: Any, **kwargs) -> Any:
        if isinstance(inputs, dict):
            return self.model(**inputs, **kwargs)
        return self.model(inputs, **kwargs)


def create_inference_runner(model: nn.Module, device: str, precision: str = "fp16", model_type: str = "base",
                            **kwargs) -> InferenceRunner:
    """inference.py:1779-1810.  "base" and "transformer" share the runner here (the paged cache is owned by the
    caller: PagedKVCache above); "diffusion" is out of scope."""
    if model_type in ("base", "transformer"):
        return BasicInferenceRunner(model, device, precision)
    raise ValueError(f"Unsupported model type: {model_type}")
