"""CPU oracle for the MI355X hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch fp32/fp64 on the host) of the reference's
algorithms for the hot path named in BASELINE.json `north_star`:
exact softmax attention / online-softmax flash attention, ring attention
(semantics A of SURVEY.md F5), FusedMLP (gelu-tanh, gelu-erf, relu, silu, swiglu),
the (o, lse) merge, LayerNorm, paged decode attention, reshape_and_cache and the
`baseline/inference.py` BasicInferenceRunner timing harness.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import this package, and only as the checker -- never as the product path.
The product (`ml-inference-optimizer_amd/mio`) never imports it and fails loudly
when the HIP extension is missing.

Pinning: the reference holds no golden vectors for this path.
The restatement is pinned by fixtures generated in the build container by importing
the reference's own runnable pieces (`kernels.mlp.fused_mlp`,
`kernels.triton.mlp_kernels.pytorch_fused_mlp`, the PyTorch ring fallback in
`kernels.triton.attention_kernels`, `kernels.triton.layernorm_kernels.pytorch_layernorm`);
see `tests/golden/make_golden.py` (committed) and `tests/golden/*.npz`.
The FA3 module itself cannot be imported (SyntaxError) and its PyTorch
fallback returns zeros (F4); for it the oracle restates the kernel math at
`kernels/triton/flash_attention_kernels.py:238-305` and is cross-checked against the
reference's own comparator `standard_attention` (`kernels/attention/flash_attention.py:1216-1229`)
and against the ring fallback with an additive -1e9 triangular mask.
"""
from .attention import (  # noqa: F401
    standard_attention,
    flash_attention_online,
    ring_attention_forward,
    attention_with_lse,
    merge_attention_states,
    paged_attention_forward,
    reshape_and_cache,
)
from .mlp import fused_mlp, gelu_tanh, ACTIVATIONS  # noqa: F401
from .rowops import layernorm, layernorm_residual  # noqa: F401
