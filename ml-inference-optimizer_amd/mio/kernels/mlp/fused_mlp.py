"""FusedMLP operator surface on the HIP kernels.

Mirrors reference kernels/mlp/fused_mlp.py: FusedMLPConfig (:14-25), FusedMLP (:28-202),
FusedMLPGeluTanh (:205-237), FusedMLPSwiGLU (:240-296), FusedMLPReLU (:299-315),
FusedTransformerMLP (:318-396), MLPConverter (:399-613) -- same constructor / forward signatures,
parameter names (fc1, fc2, fc1_gate) and error behaviour.  forward() is ONE call into
mio_fused_mlp_fwd (two MFMA GEMMs with bias/activation/gate fused in the first one's epilogue);
there is no PyTorch path: CPU tensors raise ValueError like the reference's Triton path (:89-90).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional

import torch
import torch.nn as nn

from ... import ops
from ..._nn import CastCache, ResidualStream, compute_dtype


@dataclass
class FusedMLPConfig:
    """Configuration for FusedMLP modules (fields as reference fused_mlp.py:14-25)."""
    activation_fn: str = "gelu"
    dropout_prob: float = 0.0
    use_triton: bool = True          # kept for signature compatibility; the HIP kernel is always used
    precision: str = "fp16"
    fuse_bias_gelu: bool = True
    recompute_activation: bool = False
    sequence_parallel: bool = False
    tensor_parallel: bool = False
    checkpoint_activation: bool = False


class FusedMLP(nn.Module):
    """fc2(act(fc1(x))) with act in {gelu (exact erf, :162-163), relu, silu}."""

    #: activation name handed to the kernel for config.activation_fn == "gelu"
    _gelu_kind = "gelu_erf"

    def __init__(self, hidden_size: int, intermediate_size: int, config: Optional[FusedMLPConfig] = None):
        super().__init__()
        self.config = config or FusedMLPConfig()
        self.fc1 = nn.Linear(hidden_size, intermediate_size, bias=True)
        self.fc2 = nn.Linear(intermediate_size, hidden_size, bias=True)
        self.dropout = nn.Dropout(self.config.dropout_prob) if self.config.dropout_prob > 0 else None
        self._cast = CastCache()

    def _kernel_activation(self) -> str:
        act = self.config.activation_fn
        if act == "gelu":
            return self._gelu_kind
        if act in ("relu", "silu"):
            return act
        raise ValueError(f"Unsupported activation function: {act}")

    def _gate(self, dtype):
        return None, None

    def stream_ok(self, B: int, S: int, dtype: torch.dtype, pre_norm: Optional[nn.LayerNorm]) -> bool:
        """True iff forward(...) can take (and return) the residual stream as a ResidualStream at this size (ops.gemm_ln_ok on
        both GEMMs; tanh-GELU or SwiGLU)."""
        act = self._kernel_activation()
        d, I, M = self.fc1.in_features, self.fc1.out_features, B * S
        if dtype not in (torch.float16, torch.bfloat16) or pre_norm is None or pre_norm.weight is None or ops.NO_BLOCKED_X:
            return False
        if compute_dtype(self.config.precision, torch.empty(0, dtype=dtype)) != dtype:
            return False  # the stream form runs in the stream's dtype
        if act not in ("gelu", "swiglu") or tuple(pre_norm.normalized_shape) != (d,) or self.fc2.out_features != d:
            return False
        if self.training and self.dropout is not None:
            return False
        return (ops.fused_mlp_blocked_weight_ok(M, d, I, act) and ops.gemm_ln_ok(M, I, d, act, fold_in=True)
                and ops.gemm_ln_ok(M, d, I, "none", stats_out=True))

    def _forward_stream(self, x: ResidualStream, pre_norm: nn.LayerNorm, stream_out: bool):
        """fc1 normalises the raw stream in its read-out and writes act(...) blocked; fc2 reads the residual from the blocked
        stream and writes the new stream (blocked + row statistics) or a plain [B, S, d] tensor."""
        c, act = self._cast, self._kernel_activation()
        B, S, d = x.shape
        dt, M, I = x.dtype, B * S, self.fc1.out_features
        if act == "swiglu":
            wfb, bfold, bgate = c.get_ln_folded_glu(self.fc1_gate, self.fc1, pre_norm, dt)
        else:
            (wfb, bfold), bgate = c.get_ln_folded(self.fc1, pre_norm, dt), None
        h, _ = ops.gemm_ln(x.blocked, wfb, bfold, M=M, N=I, K=d, activation=act, x_blocked=True, out_blocked=True,
                           ln_stats=x.stats, eps=pre_norm.eps, bias_gate=bgate)
        y, st = ops.gemm_ln(h, c.get_blocked(self.fc2.weight, dt), c.get(self.fc2.bias, dt), M=M, N=d, K=I, x_blocked=True,
                            residual=x.blocked, res_blocked=True, out_blocked=stream_out, stats_out=stream_out)
        return ResidualStream(y, st, (B, S, d)) if stream_out else y.view(B, S, d)

    def forward(self, hidden_states: torch.Tensor, residual: Optional[torch.Tensor] = None,
                pre_norm: Optional[nn.LayerNorm] = None, stream_out: bool = False) -> torch.Tensor:
        """pre_norm (not in the reference): a LayerNorm to apply to hidden_states first -- the pre-LN block's
        `mlp(ln(x))` in one call, which lets LayerNorm hand its output to fc1 in the blocked layout.
        hidden_states may be a ResidualStream (mio._nn; the residual is then the stream itself and the LayerNorm is folded into
        fc1's read-out, ops.gemm_ln); stream_out=True returns one.  Where stream_ok() says so."""
        if isinstance(hidden_states, ResidualStream):
            B, S, _ = hidden_states.shape
            if (residual is not None and residual is not hidden_states) or not self.stream_ok(B, S, hidden_states.dtype, pre_norm):
                raise ValueError("the ResidualStream form needs pre_norm, residual = the input itself and a size with stream_ok()")
            return self._forward_stream(hidden_states, pre_norm, stream_out)
        if stream_out:
            raise ValueError("stream_out needs a ResidualStream input (the attention sub-layer produces the first one)")
        if hidden_states.dim() != 3:
            raise ValueError(f"Expected 3D input tensor, got shape: {hidden_states.shape}")
        if not hidden_states.is_cuda:
            raise ValueError("Input tensor must be on a CUDA device for the HIP kernels")
        if self.training and self.dropout is not None:
            raise NotImplementedError("dropout between fc1 and fc2 (training) is not part of the fused inference kernel")
        act = self._kernel_activation()
        in_dtype = hidden_states.dtype
        dt = compute_dtype(self.config.precision, hidden_states)
        x = hidden_states if in_dtype == dt else hidden_states.to(dt)
        r = None if residual is None else (residual if residual.dtype == dt else residual.to(dt))
        c = self._cast
        gw, gb = self._gate(dt)
        M, d, I = x.numel() // x.shape[-1], x.shape[-1], self.fc1.weight.shape[0]
        b1 = b2 = None  # blocked weight copies, when this shape runs the kernels that take them
        if ops.fused_mlp_blocked_weight_ok(M, d, I, act):
            if gw is None:
                b1 = c.get_blocked(self.fc1.weight, dt)
            else:  # SwiGLU: gate and up rows interleaved in one blocked weight (ops.block_weight_glu)
                b1 = c.get_blocked_glu(self.fc1_gate.weight, self.fc1.weight, dt)
            b2 = c.get_blocked(self.fc2.weight, dt)
        xshape = None
        if pre_norm is not None:
            lw, lb = c.get(pre_norm.weight, dt), c.get(pre_norm.bias, dt)
            if b1 is not None and d % 32 == 0 and not ops.NO_BLOCKED_X:
                xshape = tuple(x.shape)
                x = ops.layernorm(x, lw, lb, pre_norm.eps, out_blocked=True)
            else:
                x = ops.layernorm(x, lw, lb, pre_norm.eps)
        out = ops.fused_mlp(x, c.get(self.fc1.weight, dt), c.get(self.fc1.bias, dt), c.get(self.fc2.weight, dt),
                            c.get(self.fc2.bias, dt), act, gw, gb, residual=r, fc1_blocked=b1, fc2_blocked=b2,
                            x_blocked_shape=xshape)
        return out if out.dtype == in_dtype else out.to(in_dtype)


class FusedMLPGeluTanh(FusedMLP):
    """tanh-approximated GELU (reference :205-237; == the Triton kernel's activation)."""
    _gelu_kind = "gelu"

    def __init__(self, hidden_size: int, intermediate_size: int, config: Optional[FusedMLPConfig] = None):
        config = config or FusedMLPConfig()
        config.activation_fn = "gelu"
        super().__init__(hidden_size, intermediate_size, config)


class FusedMLPSwiGLU(FusedMLP):
    """fc2(silu(fc1_gate(x)) * fc1(x))  (reference :240-296)."""

    def __init__(self, hidden_size: int, intermediate_size: int, config: Optional[FusedMLPConfig] = None):
        config = config or FusedMLPConfig()
        config.activation_fn = "swiglu"
        super().__init__(hidden_size, intermediate_size, config)
        self.fc1_gate = nn.Linear(hidden_size, intermediate_size, bias=True)

    def _kernel_activation(self) -> str:
        return "swiglu"

    def _gate(self, dtype):
        return self._cast.get(self.fc1_gate.weight, dtype), self._cast.get(self.fc1_gate.bias, dtype)


class FusedMLPReLU(FusedMLP):
    """ReLU variant (reference :299-315)."""

    def __init__(self, hidden_size: int, intermediate_size: int, config: Optional[FusedMLPConfig] = None):
        config = config or FusedMLPConfig()
        config.activation_fn = "relu"
        super().__init__(hidden_size, intermediate_size, config)


class FusedTransformerMLP(nn.Module):
    """Drop-in replacement for transformer MLP blocks (reference :318-396)."""

    def __init__(self, hidden_size: int, intermediate_size: int, activation_fn: str = "gelu",
                 config: Optional[FusedMLPConfig] = None):
        super().__init__()
        self.config = config or FusedMLPConfig()
        self.config.activation_fn = activation_fn
        if activation_fn == "gelu":
            self.mlp = FusedMLPGeluTanh(hidden_size, intermediate_size, self.config)
        elif activation_fn == "swiglu":
            self.mlp = FusedMLPSwiGLU(hidden_size, intermediate_size, self.config)
        elif activation_fn == "relu":
            self.mlp = FusedMLPReLU(hidden_size, intermediate_size, self.config)
        else:
            self.mlp = FusedMLP(hidden_size, intermediate_size, self.config)

    def stream_ok(self, B: int, S: int, dtype: torch.dtype, pre_norm: Optional[nn.LayerNorm]) -> bool:
        return self.mlp.stream_ok(B, S, dtype, pre_norm)

    def forward(self, hidden_states: torch.Tensor, residual: Optional[torch.Tensor] = None,
                pre_norm: Optional[nn.LayerNorm] = None, stream_out: bool = False) -> torch.Tensor:
        if isinstance(hidden_states, ResidualStream) or stream_out:
            return self.mlp(hidden_states, residual, pre_norm, stream_out)
        if pre_norm is not None:
            return self.mlp(hidden_states, residual, pre_norm)
        return self.mlp(hidden_states, residual) if residual is not None else self.mlp(hidden_states)

    def load_from_standard_mlp(self, state_dict: Dict[str, torch.Tensor], prefix: str = "") -> None:
        """Key mapping of reference :362-396 (dense/output.dense, gate_proj/up_proj/down_proj)."""
        key_mapping = {
            f"{prefix}dense.weight": "mlp.fc1.weight",
            f"{prefix}dense.bias": "mlp.fc1.bias",
            f"{prefix}output.dense.weight": "mlp.fc2.weight",
            f"{prefix}output.dense.bias": "mlp.fc2.bias",
        }
        if isinstance(self.mlp, FusedMLPSwiGLU) and f"{prefix}gate_proj.weight" in state_dict:
            key_mapping.update({
                f"{prefix}gate_proj.weight": "mlp.fc1_gate.weight",
                f"{prefix}gate_proj.bias": "mlp.fc1_gate.bias",
                f"{prefix}up_proj.weight": "mlp.fc1.weight",
                f"{prefix}up_proj.bias": "mlp.fc1.bias",
                # the reference forgets down_proj (its LLaMA conversions keep a random fc2); map it
                f"{prefix}down_proj.weight": "mlp.fc2.weight",
                f"{prefix}down_proj.bias": "mlp.fc2.bias",
            })
        own = self.state_dict()
        with torch.no_grad():
            for src, dst in key_mapping.items():
                if src in state_dict and state_dict[src] is not None and dst in own:
                    own[dst].copy_(state_dict[src])
        self.load_state_dict(own)


def _weight_out_in(lin: nn.Module) -> torch.Tensor:
    """nn.Linear stores [out, in]; HF GPT-2's Conv1D stores [in, out] (SURVEY a4/a8 note)."""
    w = lin.weight
    if type(lin).__name__ == "Conv1D":
        return w.t()
    return w


def _features(lin: nn.Module):
    if type(lin).__name__ == "Conv1D":
        return lin.weight.shape[0], lin.weight.shape[1]
    return lin.in_features, lin.out_features


class MLPConverter:
    """Scan a model and replace MLP blocks with FusedTransformerMLP (reference :399-613).
    Detects the reference's three patterns (HF dense/output.dense, torch linear1/linear2,
    LLaMA gate/up/down) plus HF GPT-2's c_fc/c_proj Conv1D block, which the reference misses."""

    def __init__(self, config: Optional[FusedMLPConfig] = None):
        self.config = config or FusedMLPConfig()
        self.activation_map = {"gelu": "gelu", "relu": "relu", "silu": "silu", "swish": "silu",
                               "swiglu": "swiglu", "gelu_new": "gelu"}

    @staticmethod
    def _act_of(mod: Optional[nn.Module], default: str = "gelu") -> str:
        if mod is None:
            return default
        name = mod.__class__.__name__.lower()
        if "gelu" in name:
            return "gelu"
        if "relu" in name:
            return "relu"
        if "silu" in name or "swish" in name:
            return "silu"
        return default

    def _detect_mlp_type(self, module: nn.Module) -> Optional[Dict[str, Any]]:
        if hasattr(module, "dense") and hasattr(module, "output") and hasattr(module.output, "dense"):
            i, o = _features(module.dense)
            return {"hidden_size": i, "intermediate_size": o,
                    "activation_fn": self._act_of(getattr(module, "act", None)), "pattern": "huggingface"}
        if hasattr(module, "linear1") and hasattr(module, "linear2"):
            i, o = _features(module.linear1)
            act = "gelu"
            for attr_name, attr in module.named_children():
                if "activation" in attr_name.lower() or attr_name.lower() == "act":
                    act = self._act_of(attr, act)
            return {"hidden_size": i, "intermediate_size": o, "activation_fn": act, "pattern": "pytorch"}
        if hasattr(module, "gate_proj") and hasattr(module, "up_proj") and hasattr(module, "down_proj"):
            i, o = _features(module.gate_proj)
            return {"hidden_size": i, "intermediate_size": o, "activation_fn": "swiglu", "pattern": "llama"}
        if hasattr(module, "c_fc") and hasattr(module, "c_proj"):
            i, o = _features(module.c_fc)
            return {"hidden_size": i, "intermediate_size": o,
                    "activation_fn": self._act_of(getattr(module, "act", None)), "pattern": "gpt2"}
        return None

    def _create_fused_mlp(self, mlp_params: Dict[str, Any]) -> FusedTransformerMLP:
        act = self.activation_map.get(mlp_params["activation_fn"], mlp_params["activation_fn"])
        import copy
        return FusedTransformerMLP(hidden_size=mlp_params["hidden_size"],
                                   intermediate_size=mlp_params["intermediate_size"], activation_fn=act,
                                   config=copy.copy(self.config))

    def _copy_weights(self, fused_mlp: FusedTransformerMLP, original_mlp: nn.Module, mlp_params: Dict[str, Any]) -> None:
        def wb(lin, wkey, bkey, sd):
            sd[wkey] = _weight_out_in(lin)
            b = getattr(lin, "bias", None)
            sd[bkey] = b if b is not None else torch.zeros(sd[wkey].shape[0], dtype=sd[wkey].dtype,
                                                            device=sd[wkey].device)
        sd: Dict[str, torch.Tensor] = {}
        pattern = mlp_params["pattern"]
        if pattern == "huggingface":
            wb(original_mlp.dense, "dense.weight", "dense.bias", sd)
            wb(original_mlp.output.dense, "output.dense.weight", "output.dense.bias", sd)
        elif pattern == "pytorch":
            wb(original_mlp.linear1, "dense.weight", "dense.bias", sd)
            wb(original_mlp.linear2, "output.dense.weight", "output.dense.bias", sd)
        elif pattern == "gpt2":
            wb(original_mlp.c_fc, "dense.weight", "dense.bias", sd)
            wb(original_mlp.c_proj, "output.dense.weight", "output.dense.bias", sd)
        elif pattern == "llama":
            wb(original_mlp.gate_proj, "gate_proj.weight", "gate_proj.bias", sd)
            wb(original_mlp.up_proj, "up_proj.weight", "up_proj.bias", sd)
            wb(original_mlp.down_proj, "down_proj.weight", "down_proj.bias", sd)
        fused_mlp.load_from_standard_mlp(sd)

    def convert_model(self, model: nn.Module, target_class_names: Optional[list] = None) -> nn.Module:
        if target_class_names is None:
            target_class_names = ["MLP", "FFN", "FeedForward", "MLPBlock", "GELU_MLP", "SwiGLU",
                                  "FeedForwardNetwork", "PositionwiseFeedForward"]
        self.replacements = 0

        def _process(module: nn.Module, parent: Optional[nn.Module] = None, name: str = ""):
            if isinstance(module, (FusedTransformerMLP, FusedMLP)):
                return
            if parent is not None and any(t in module.__class__.__name__ for t in target_class_names):
                params = self._detect_mlp_type(module)
                if params:
                    ref_p = next(module.parameters(), None)
                    fused = self._create_fused_mlp(params)
                    if ref_p is not None:
                        fused = fused.to(device=ref_p.device, dtype=ref_p.dtype)
                    self._copy_weights(fused, module, params)
                    fused.train(module.training)
                    setattr(parent, name, fused)
                    self.replacements += 1
                    return
            for child_name, child in list(module.named_children()):
                _process(child, module, child_name)

        _process(model)
        return model
