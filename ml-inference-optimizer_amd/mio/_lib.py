"""ctypes binding of libmio_hip.so (the C ABI declared in include/mio_hip.h).

There is no fallback: if the shared library is missing or a symbol is absent, importing this
module raises.  PyTorch is only used by the callers for device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch must be loaded first: it ships its own HIP runtime (torch/lib/libamdhip64.so, SONAME
# libamdhip64.so.7).  With it already mapped, libmio_hip.so's NEEDED libamdhip64.so.7 binds to that same
# runtime, so torch's streams/events/allocations and our launches live in ONE HIP runtime.  (Loaded the other
# way round, /opt/rocm's copy would be pulled in first and the process would hold two runtimes.)
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MIO_LIB_DBG=1 (tools/ only): the diagnostic build (`make dbg`), which carries the A/B switches and the in-kernel
# stamp instantiations; the product library reads no environment variable and has neither.
LIB_PATH = os.path.join(_HERE, "libmio_hip_dbg.so" if os.environ.get("MIO_LIB_DBG") == "1" else "libmio_hip.so")

MIO_BF16, MIO_FP16 = 0, 1
ACT_NONE, ACT_GELU_TANH, ACT_GELU_ERF, ACT_RELU, ACT_SILU, ACT_SWIGLU = range(6)
MASK_NONE, MASK_KEEP_U8, MASK_ADD_F32 = range(3)

# every symbol include/mio_hip.h declares
EXPORTS = (
    "mio_version",
    "mio_last_error",
    "mio_fa3_fwd",
    "mio_fa3_k_prescaled_ok",
    "mio_fa3_o_blocked_ok",
    "mio_attn_merge",
    "mio_gemm_bias_act",
    "mio_fused_mlp_workspace_bytes",
    "mio_fused_mlp_fwd",
    "mio_weight_blocked_bytes",
    "mio_weight_block",
    "mio_gemm_blocked_weight_ok",
    "mio_gemm_bias_act_bw",
    "mio_gemm_col_scale_ok",
    "mio_gemm_bias_act_bw_cs",
    "mio_fused_mlp_blocked_weight_ok",
    "mio_fused_mlp_fwd_bw",
    "mio_weight_blocked_glu_bytes",
    "mio_weight_block_glu",
    "mio_fused_mlp_glu_fwd_bw",
    "mio_layernorm_fwd_bx",
    "mio_ln_stats_bytes",
    "mio_gemm_ln_ok",
    "mio_ln_fold_weight",
    "mio_gemm_ln_bw",
    "mio_ln_stats_reduce",
    "mio_layernorm_fwd",
    "mio_fa3_decode_workspace_bytes",
    "mio_fa3_decode_paged",
    "mio_reshape_and_cache",
)


class FaParams(C.Structure):
    """mio_fa3_fwd_params_t"""

    _fields_ = [
        ("q", C.c_void_p),
        ("k", C.c_void_p),
        ("v", C.c_void_p),
        ("o", C.c_void_p),
        ("lse", C.c_void_p),
        ("o_acc", C.c_void_p),
        ("mask", C.c_void_p),
        ("q_stride", C.c_int64 * 3),
        ("k_stride", C.c_int64 * 3),
        ("v_stride", C.c_int64 * 3),
        ("o_stride", C.c_int64 * 3),
        ("mask_stride", C.c_int64 * 4),
        ("B", C.c_int32),
        ("Sq", C.c_int32),
        ("Sk", C.c_int32),
        ("H", C.c_int32),
        ("Hkv", C.c_int32),
        ("D", C.c_int32),
        ("dtype", C.c_int32),
        ("causal", C.c_int32),
        ("mask_kind", C.c_int32),
        ("carry_in", C.c_int32),
        ("q_offset", C.c_int32),
        ("k_offset", C.c_int32),
        ("softmax_scale", C.c_float),
        ("k_prescaled", C.c_int32),
        ("o_blocked", C.c_int32),
    ]


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `make -C ml-inference-optimizer_amd/csrc` "
            "(or __graft_entry__.build()). There is no CPU or PyTorch fallback for this path."
        )
    lib = C.CDLL(LIB_PATH)
    missing = [s for s in EXPORTS if not hasattr(lib, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} lacks symbols {missing}; rebuild the extension")
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    lib.mio_version.restype = i32
    lib.mio_last_error.restype = C.c_char_p
    lib.mio_fa3_fwd.argtypes = [C.POINTER(FaParams), vp]
    lib.mio_fa3_fwd.restype = i32
    lib.mio_fa3_k_prescaled_ok.argtypes = [C.POINTER(FaParams)]
    lib.mio_fa3_k_prescaled_ok.restype = i32
    lib.mio_fa3_o_blocked_ok.argtypes = [C.POINTER(FaParams)]
    lib.mio_fa3_o_blocked_ok.restype = i32
    lib.mio_attn_merge.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.mio_attn_merge.restype = i32
    lib.mio_gemm_bias_act.argtypes = [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, i64, i64, i64, i32, i32, vp]
    lib.mio_gemm_bias_act.restype = i32
    lib.mio_fused_mlp_workspace_bytes.argtypes = [i64, i32, i32, i32]
    lib.mio_fused_mlp_workspace_bytes.restype = C.c_size_t
    lib.mio_fused_mlp_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.mio_fused_mlp_fwd.restype = i32
    lib.mio_weight_blocked_bytes.argtypes = [i32, i32]
    lib.mio_weight_blocked_bytes.restype = C.c_size_t
    lib.mio_weight_block.argtypes = [vp, i64, vp, i32, i32, i32, vp]
    lib.mio_weight_block.restype = i32
    lib.mio_gemm_blocked_weight_ok.argtypes = [i64, i32, i32, i32]
    lib.mio_gemm_blocked_weight_ok.restype = i32
    lib.mio_gemm_bias_act_bw.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i64, i64, i64, i32, i32, i32, vp]
    lib.mio_gemm_bias_act_bw.restype = i32
    lib.mio_gemm_col_scale_ok.argtypes = [i64, i32, i32, i32]
    lib.mio_gemm_col_scale_ok.restype = i32
    lib.mio_gemm_bias_act_bw_cs.argtypes = [vp, vp, vp, vp, i64, i32, i32, i64, i64, i32, i32, i32, i32, i32, f32, vp]
    lib.mio_gemm_bias_act_bw_cs.restype = i32
    lib.mio_fused_mlp_blocked_weight_ok.argtypes = [i64, i32, i32, i32]
    lib.mio_fused_mlp_blocked_weight_ok.restype = i32
    lib.mio_fused_mlp_fwd_bw.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, i32, vp]
    lib.mio_fused_mlp_fwd_bw.restype = i32
    lib.mio_weight_blocked_glu_bytes.argtypes = [i32, i32]
    lib.mio_weight_blocked_glu_bytes.restype = C.c_size_t
    lib.mio_weight_block_glu.argtypes = [vp, vp, i64, vp, i32, i32, i32, vp]
    lib.mio_weight_block_glu.restype = i32
    lib.mio_fused_mlp_glu_fwd_bw.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, vp]
    lib.mio_fused_mlp_glu_fwd_bw.restype = i32
    lib.mio_ln_stats_bytes.argtypes = [i64, i32]
    lib.mio_ln_stats_bytes.restype = C.c_size_t
    lib.mio_gemm_ln_ok.argtypes = [i64, i32, i32, i32, i32, i32]
    lib.mio_gemm_ln_ok.restype = i32
    lib.mio_ln_fold_weight.argtypes = [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, vp]
    lib.mio_ln_fold_weight.restype = i32
    lib.mio_gemm_ln_bw.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i64, i64, i64, i32, i32, i32, vp, i32, f32, vp, i32, i32, f32, vp]
    lib.mio_ln_stats_reduce.argtypes = [vp, i32, vp, i32, i64, vp]
    lib.mio_ln_stats_reduce.restype = i32
    lib.mio_gemm_ln_bw.restype = i32
    lib.mio_layernorm_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp]
    lib.mio_layernorm_fwd.restype = i32
    lib.mio_layernorm_fwd_bx.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp]
    lib.mio_layernorm_fwd_bx.restype = i32
    lib.mio_fa3_decode_workspace_bytes.argtypes = [i32, i32, i32, i32, i32]
    lib.mio_fa3_decode_workspace_bytes.restype = C.c_size_t
    lib.mio_fa3_decode_paged.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64), i32, i32, i32, i32,
                                         i32, i32, i32, i32, i32, i32, f32, i32, vp, vp]
    lib.mio_fa3_decode_paged.restype = i32
    lib.mio_reshape_and_cache.argtypes = [vp, vp, vp, vp, vp, vp, C.POINTER(i64), C.POINTER(i64), i32, i32, i32, i32,
                                          i32, i32, i32, i32, vp]
    lib.mio_reshape_and_cache.restype = i32
    return lib


lib = _load()


def check(rc: int) -> None:
    """Non-zero C return -> RuntimeError(mio_last_error()); no silent fallback."""
    if rc != 0:
        raise RuntimeError(lib.mio_last_error().decode("utf-8", "replace"))
