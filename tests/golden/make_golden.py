#!/usr/bin/env python3
"""Generate golden vectors by importing the runnable pieces of the reference.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

What is imported (all plain-PyTorch code paths of the reference, executed on CPU in fp32):
  * kernels.mlp.fused_mlp.FusedTransformerMLP             (fused_mlp.py:318-396)
  * kernels.triton.mlp_kernels.pytorch_fused_mlp          (mlp_kernels.py:759-803)
  * kernels.triton.attention_kernels.triton_ring_attention_forward -- the PyTorch ring
    fallback (attention_kernels.py:1520-1591); it is only defined when `import triton` fails,
    so triton is hidden with sys.modules['triton'] = None before the import
  * kernels.triton.layernorm_kernels.pytorch_layernorm    (layernorm_kernels.py:279-311)
The FA3 module is unimportable (SyntaxError) and is therefore not a generator; causal
attention goldens come from the ring fallback driven with an additive -1e9 triangular mask,
which is the reference's own masking rule (flash_attention_kernels.py:254).

Outputs are data only (inputs + expected outputs, fp32) in tests/golden/*.npz.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _mlp_goldens():
    sys.path.insert(0, REF)
    from kernels.mlp.fused_mlp import FusedTransformerMLP, FusedMLP, FusedMLPConfig  # noqa

    out = {}
    for name, act, d, I, B, S in [
        ("gelu", "gelu", 64, 256, 2, 75),
        ("swiglu", "swiglu", 64, 192, 2, 75),
        ("relu", "relu", 96, 160, 1, 130),
        ("silu", "silu", 64, 256, 3, 33),
    ]:
        torch.manual_seed(1234 + len(out))
        m = FusedTransformerMLP(d, I, activation_fn=act, config=FusedMLPConfig(use_triton=False)).eval()
        x = torch.randn(B, S, d)
        with torch.no_grad():
            y = m(x)
        out[f"{name}_x"] = x.numpy()
        out[f"{name}_y"] = y.numpy()
        for k, v in m.state_dict().items():
            out[f"{name}_{k.replace('.', '_')}"] = v.numpy()
    # exact-erf GELU: the FusedMLP base class with activation_fn == "gelu" (fused_mlp.py:162-163)
    torch.manual_seed(99)
    m = FusedMLP(64, 256, FusedMLPConfig(activation_fn="gelu", use_triton=False)).eval()
    x = torch.randn(2, 50, 64)
    with torch.no_grad():
        y = m(x)
    out["gelu_erf_x"], out["gelu_erf_y"] = x.numpy(), y.numpy()
    for k, v in m.state_dict().items():
        out[f"gelu_erf_{k.replace('.', '_')}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "fused_mlp_modules.npz"), **out)

    from kernels.triton.mlp_kernels import pytorch_fused_mlp

    out = {}
    for act in ("gelu", "relu", "swiglu"):
        torch.manual_seed(7 + len(out))
        d, I, B, S = 64, 128, 2, 41
        x = torch.randn(B, S, d)
        w1, b1 = torch.randn(I, d) * 0.1, torch.randn(I) * 0.1
        w2, b2 = torch.randn(d, I) * 0.1, torch.randn(d) * 0.1
        wg, bg = torch.randn(I, d) * 0.1, torch.randn(I) * 0.1
        y = pytorch_fused_mlp(x, w1, b1, w2, b2, act, wg if act == "swiglu" else None, bg if act == "swiglu" else None)
        for k, v in dict(x=x, w1=w1, b1=b1, w2=w2, b2=b2, wg=wg, bg=bg, y=y).items():
            out[f"{act}_{k}"] = v.numpy()
    np.savez_compressed(os.path.join(OUT, "fused_mlp_functional.npz"), **out)


def _ring_goldens():
    sys.modules["triton"] = None  # make `import triton` fail -> the PyTorch fallback is defined
    sys.modules.pop("kernels.triton.attention_kernels", None)
    from kernels.triton.attention_kernels import triton_ring_attention_forward

    out = {}
    for name, B, H, Sq, Sk, D, maskkind in [
        ("d64_nomask", 2, 2, 300, 300, 64, "none"),
        ("d64_additive", 1, 2, 200, 200, 64, "additive"),
        ("d64_causal", 1, 2, 260, 260, 64, "causal"),
        ("d80_cross", 1, 2, 130, 333, 80, "none"),
        ("d128_causal", 1, 1, 192, 192, 128, "causal"),
        ("d64_padding", 2, 1, 130, 130, 64, "padding"),
    ]:
        torch.manual_seed(4321 + len(out))
        q = torch.randn(B, H, Sq, D)
        k = torch.randn(B, H, Sk, D)
        v = torch.randn(B, H, Sk, D)
        mask = None
        if maskkind == "additive":
            mask = torch.randn(B, 1, Sq, Sk) * 2.0
        elif maskkind == "causal":
            mask = torch.triu(torch.full((Sq, Sk), -1e9), diagonal=1)[None, None].expand(B, 1, Sq, Sk).contiguous()
        elif maskkind == "padding":
            keep = torch.ones(B, Sk)
            keep[0, 100:] = 0
            keep[1, 77:] = 0
            mask = ((1.0 - keep) * -1e9)[:, None, None, :].expand(B, 1, Sq, Sk).contiguous()
            out[f"{name}_keep"] = keep.numpy()
        o = triton_ring_attention_forward(q, k, v, mask)
        out[f"{name}_q"], out[f"{name}_k"], out[f"{name}_v"] = q.numpy(), k.numpy(), v.numpy()
        if mask is not None and maskkind == "additive":
            out[f"{name}_mask"] = mask.numpy()
        out[f"{name}_o"] = o.numpy()
    np.savez_compressed(os.path.join(OUT, "ring_attention_fallback.npz"), **out)
    del sys.modules["triton"]


def _ln_goldens():
    sys.modules["triton"] = None
    sys.modules.pop("kernels.triton.layernorm_kernels", None)
    try:
        from kernels.triton.layernorm_kernels import pytorch_layernorm
    finally:
        del sys.modules["triton"]
    out = {}
    torch.manual_seed(5)
    x = torch.randn(2, 9, 1024) * 2 + 0.3
    r = torch.randn(2, 9, 1024)
    w, b = torch.randn(1024), torch.randn(1024)
    out["x"], out["r"], out["w"], out["b"] = x.numpy(), r.numpy(), w.numpy(), b.numpy()
    out["y"] = pytorch_layernorm(x, w, b, 1e-5).numpy()
    out["y_res"] = pytorch_layernorm(x, w, b, 1e-5, residual=r, residual_alpha=0.5).numpy()
    np.savez_compressed(os.path.join(OUT, "layernorm.npz"), **out)


if __name__ == "__main__":
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    _mlp_goldens()
    _ring_goldens()
    try:
        _ln_goldens()
    except Exception as e:  # layernorm_kernels imports triton unconditionally
        print("layernorm golden skipped:", repr(e))
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(OUT, f)))
