"""CPU-only tests: the C-ABI library loads and exports every symbol of include/mio_hip.h, the operator
surface mirrors the reference's names/errors, converters copy weights, host-side helpers."""
import os
import re
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_exports_match_header():
    from mio import _lib
    hdr = open(os.path.join(ROOT, "include", "mio_hip.h")).read()
    declared = set(re.findall(r"\b(mio_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(_lib.lib, name), name
    assert _lib.lib.mio_version() == 105
    assert _lib.lib.mio_last_error() is not None  # callable without a GPU


def test_layernorm_fold_eligibility_without_gpu():
    """mio_gemm_ln_ok / mio_ln_stats_bytes are host functions: the shapes of the benchmark stacks take the folded kernels, the
    shapes outside their limits do not (K % 256, more than 8 statistic slots, too few tiles, activations without an instantiation),
    and mio_gemm_ln_bw refuses bad argument combinations before any launch."""
    from mio import _lib
    lib = _lib.lib
    NONE, GELU_TANH, RELU, SWIGLU = _lib.ACT_NONE, 1, _lib.ACT_RELU, _lib.ACT_SWIGLU
    M = 8 * 4096
    assert lib.mio_gemm_ln_ok(M, 3072, 1024, NONE, 1, 0) == 1          # C2 QKV behind ln_1
    assert lib.mio_gemm_ln_ok(M, 4096, 1024, GELU_TANH, 1, 0) == 1     # C2 fc1 behind ln_2
    assert lib.mio_gemm_ln_ok(M, 1024, 1024, NONE, 0, 1) == 1 and lib.mio_gemm_ln_ok(M, 1024, 4096, NONE, 0, 1) == 1  # out-proj, fc2
    assert lib.mio_gemm_ln_ok(M, 5120, 1280, GELU_TANH, 1, 0) == 1 and lib.mio_gemm_ln_ok(M, 1280, 5120, NONE, 0, 1) == 1  # C5
    assert lib.mio_gemm_ln_ok(M, 4096, 1024, SWIGLU, 1, 0) == 1 and lib.mio_gemm_ln_ok(M, 4096, 1024, SWIGLU, 0, 1) == 0
    assert lib.mio_gemm_ln_ok(M, 3072, 1024 + 64, NONE, 1, 0) == 0     # K % 256
    assert lib.mio_gemm_ln_ok(M, 3072, 4096, NONE, 1, 0) == 1          # 16 statistic slots: through mio_ln_stats_reduce
    assert lib.mio_gemm_ln_ok(M, 1024 + 128, 1024, NONE, 0, 1) == 0    # N % 256
    assert lib.mio_gemm_ln_ok(1024, 1024, 1024, NONE, 0, 1) == 0       # 16 tiles: not a 256-tile launch
    assert lib.mio_gemm_ln_ok(M, 3072, 1024, RELU, 1, 0) == 0
    assert lib.mio_gemm_ln_ok(M, 1024, 1024, GELU_TANH, 0, 1) == 0     # statistics come from the plain residual epilogue
    assert lib.mio_ln_stats_bytes(M, 1024) == 4 * M * 8 and lib.mio_ln_stats_bytes(16500, 1280) == 5 * 16640 * 8
    one = 16  # any non-null 16-byte aligned "pointer": validation returns before anything is dereferenced
    assert lib.mio_gemm_ln_bw(one, one, None, None, None, one, M, 1024, 1024, 1024, 1024, 0, NONE, 0, 0, None, 0, 1e-5, one, 0, 0, 1.0, None) != 0
    assert b"residual" in lib.mio_last_error()                         # stats_out without a residual
    assert lib.mio_gemm_ln_bw(one, one, None, None, one, one, M, 3072, 1024, 1024, 3072, 3072, NONE, 0, 0, one, 0, 1e-5, None, 0, 0, 1.0, None) != 0
    assert b"no residual" in lib.mio_last_error()                      # consumer form with a residual
    assert lib.mio_gemm_ln_bw(one, one, None, one, None, one, M, 3072, 1024, 1024, 3072, 0, NONE, 0, 0, None, 0, 1e-5, None, 0, 0, 1.0, None) != 0
    assert b"bias_gate" in lib.mio_last_error()
    assert lib.mio_gemm_ln_bw(one, one, None, None, None, one, M, 3072, 1024, 1024, 3072, 0, NONE, 0, 8, None, 0, 1e-5, None, 0, 0, 1.0, None) != 0
    assert b"flag" in lib.mio_last_error()
    assert lib.mio_gemm_ln_bw(one, one, None, None, None, one, M, 3072, 4096, 4096, 3072, 0, NONE, 0, 0, one, 0, 1e-5, None, 0, 0, 1.0, None) != 0
    assert b"mio_ln_stats_reduce" in lib.mio_last_error()               # 16 slots handed over unreduced
    assert lib.mio_ln_fold_weight(one, 16384, one, None, None, one, one, 1024, 16384, 0, None) != 0 and b"8192" in lib.mio_last_error()
    assert lib.mio_ln_stats_reduce(one, 16, one, 5, M, None) != 0


def test_cabi_argument_errors_without_gpu():
    """Validation happens before any launch, so bad arguments are reportable without a device."""
    import ctypes as C
    from mio import _lib
    p = _lib.FaParams()
    assert _lib.lib.mio_fa3_fwd(C.byref(p), None) != 0
    assert b"non-null" in _lib.lib.mio_last_error()
    assert _lib.lib.mio_gemm_bias_act(None, None, None, None, None, None, None, 1, 8, 8, 8, 8, 8, 8, 0, 0, None) != 0
    with pytest.raises(RuntimeError):
        _lib.check(-1)


def test_no_cpu_fallback():
    from mio import ops
    from mio.kernels.mlp import FusedTransformerMLP
    from mio.kernels.attention import FlashAttention3, FlashSelfAttention
    x = torch.randn(1, 4, 64)
    with pytest.raises(ValueError):
        FusedTransformerMLP(64, 128)(x)
    with pytest.raises(ValueError):
        FlashSelfAttention(64, 4)(x)
    q = torch.randn(1, 4, 2, 32)
    with pytest.raises(ValueError):
        FlashAttention3()(q, q, q)
    with pytest.raises(ValueError):
        ops.flash_attention(q, q, q)
    with pytest.raises(ValueError):
        FlashAttention3()(q[0], q, q)  # rank check first, like flash_attention.py:167


def test_configs_mirror_reference():
    from mio.kernels.attention import FlashAttentionConfig
    from mio.kernels.mlp import FusedMLPConfig
    from mio.parallelism import SequenceParallelConfig, TensorParallelConfig
    c = FlashAttentionConfig()
    assert (c.block_size, c.causal, c.softmax_scale, c.dropout_p, c.precision) == (128, False, None, 0.0, "fp16")
    with pytest.raises(ValueError):
        FlashAttentionConfig(precision="int8")
    with pytest.raises(RuntimeError):
        FlashAttentionConfig(precision="fp8")
    m = FusedMLPConfig()
    assert (m.activation_fn, m.dropout_prob, m.precision) == ("gelu", 0.0, "fp16")
    with pytest.raises(ValueError):
        SequenceParallelConfig(world_size=4, sp_size=3)
    with pytest.raises(ValueError):
        SequenceParallelConfig(attention_handling="striped")
    t = TensorParallelConfig(world_size=8, tp_size=2)
    assert t.dp_size == 4
    from mio.kernels.attention import FlashAttentionLayer
    with pytest.raises(ValueError):
        FlashAttentionLayer(100, 3)


def test_module_parameter_names():
    from mio.kernels.attention import FlashAttentionLayer, FlashSelfAttention
    from mio.kernels.mlp import FusedTransformerMLP
    assert set(dict(FlashAttentionLayer(64, 4, num_kv_heads=2).named_parameters())) == {
        f"{p}.{w}" for p in ("q_proj", "k_proj", "v_proj", "o_proj") for w in ("weight", "bias")}
    fs = FlashSelfAttention(64, 4, num_kv_heads=2)
    assert fs.qkv_proj.weight.shape == (64 + 2 * 2 * 16, 64)
    sw = FusedTransformerMLP(64, 128, "swiglu")
    assert {"mlp.fc1.weight", "mlp.fc1_gate.weight", "mlp.fc2.weight"} <= set(dict(sw.named_parameters()))
    assert type(FusedTransformerMLP(64, 128, "gelu").mlp).__name__ == "FusedMLPGeluTanh"
    assert type(FusedTransformerMLP(64, 128, "relu").mlp).__name__ == "FusedMLPReLU"
    assert type(FusedTransformerMLP(64, 128, "silu").mlp).__name__ == "FusedMLP"


def test_converters_copy_weights():
    from oracle.baseline_runner import PlainGPT2Stack
    from ml_inference_optimizer import Optimizer
    from mio.kernels.attention import FlashAttentionLayer
    from mio.kernels.mlp import FusedTransformerMLP
    model = PlainGPT2Stack(64, 4, 2, seed=3)
    ref = {k: v.clone() for k, v in model.state_dict().items()}
    opt = Optimizer(model)
    prof = opt.profile()
    assert prof["attention_modules"] == 2 and prof["mlp_modules"] == 2
    out = opt.optimize(use_flash_attention=True, use_fused_mlp=True, tensor_parallel_size=1, causal=True)
    blk = out.h[0]
    assert isinstance(blk.attn, FlashAttentionLayer) and isinstance(blk.mlp, FusedTransformerMLP)
    assert blk.attn.config.causal is True
    assert torch.equal(blk.attn.q_proj.weight, ref["h.0.attn.q_proj.weight"])
    assert torch.equal(blk.attn.o_proj.bias, ref["h.0.attn.o_proj.bias"])
    assert torch.equal(blk.mlp.mlp.fc1.weight, ref["h.0.mlp.linear1.weight"])
    assert torch.equal(blk.mlp.mlp.fc2.weight, ref["h.0.mlp.linear2.weight"])
    assert type(blk.mlp.mlp).__name__ == "FusedMLPGeluTanh"


def test_converter_llama_and_gpt2_layouts():
    from mio.kernels.mlp import MLPConverter
    from mio.kernels.attention import ModelConverter, FlashSelfAttention

    class LlamaMLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_proj = nn.Linear(32, 64, bias=False)
            self.up_proj = nn.Linear(32, 64, bias=False)
            self.down_proj = nn.Linear(64, 32, bias=False)

    class Conv1D(nn.Module):  # HF GPT-2 layout: weight [in, out]
        def __init__(self, nf, nx):
            super().__init__()
            self.weight = nn.Parameter(torch.randn(nx, nf))
            self.bias = nn.Parameter(torch.randn(nf))

    class GPT2MLP(nn.Module):
        def __init__(self):
            super().__init__()
            self.c_fc, self.c_proj, self.act = Conv1D(64, 32), Conv1D(32, 64), nn.GELU()

    class GPT2Attention(nn.Module):
        def __init__(self):
            super().__init__()
            self.c_attn, self.c_proj = Conv1D(96, 32), Conv1D(32, 32)
            self.num_heads, self.embed_dim = 4, 32

    class Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.attn = LlamaMLP(), GPT2MLP(), GPT2Attention()

    h = Holder()
    src = {k: v.clone() for k, v in h.state_dict().items()}
    MLPConverter().convert_model(h)
    assert torch.equal(h.a.mlp.fc1_gate.weight, src["a.gate_proj.weight"])
    assert torch.equal(h.a.mlp.fc1.weight, src["a.up_proj.weight"])
    assert torch.equal(h.a.mlp.fc2.weight, src["a.down_proj.weight"])
    assert h.a.mlp.fc2.bias.abs().max() == 0
    assert torch.equal(h.b.mlp.fc1.weight, src["b.c_fc.weight"].t())
    assert torch.equal(h.b.mlp.fc2.bias, src["b.c_proj.bias"])
    ModelConverter().convert_model(h)
    assert isinstance(h.attn, FlashSelfAttention) and h.attn.num_attention_heads == 4
    assert torch.equal(h.attn.qkv_proj.weight, src["attn.c_attn.weight"].t())
    assert torch.equal(h.attn.o_proj.weight, src["attn.c_proj.weight"].t())


def test_sequence_helpers():
    from mio.parallelism import partition_sequence, gather_sequence, zigzag_shard, zigzag_unshard
    x = torch.arange(2 * 16 * 3).view(2, 16, 3)
    parts = partition_sequence(x, 4)
    assert len(parts) == 4 and torch.equal(gather_sequence(parts), x)
    with pytest.raises(ValueError):
        partition_sequence(x, 5)
    shards = [zigzag_shard(x, r, 4) for r in range(4)]
    assert torch.equal(shards[0][:, :2], x[:, 0:2]) and torch.equal(shards[0][:, 2:], x[:, 14:16])
    assert torch.equal(zigzag_unshard(shards, 4), x)


def test_comm_wrappers_single_process():
    from mio.parallelism import all_reduce, all_gather, reduce_scatter, ring_exchange, get_rank, get_world_size
    t = torch.ones(3)
    assert get_rank() == 0 and get_world_size() == 1
    assert all_reduce(t) is t and all_gather(t) is t and reduce_scatter(t) is t
    assert ring_exchange(t, None)[0] is t


def test_fusion_registry_host_logic():
    """FusionRegistry API of baseline/inference.py:26-215: pattern matching, non-overlapping candidates, replacement
    inside nn.Sequential and attribute-style parents, weights carried over (module construction is host work)."""
    from mio.baseline.inference import FusionPattern, FusionRegistry, fusion_registry
    from mio.kernels.mlp.fused_mlp import FusedMLP, FusedMLPGeluTanh, FusedMLPReLU
    assert [p.name for p in fusion_registry.patterns] == ["linear_gelu_linear", "linear_relu_linear"]
    seq = nn.Sequential(nn.LayerNorm(8), nn.Linear(8, 32), nn.GELU(approximate="tanh"), nn.Linear(32, 8),
                        nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 8), nn.Linear(8, 24), nn.GELU(), nn.Linear(24, 6))
    fused = fusion_registry.fuse_modules(seq)
    kinds = [type(m).__name__ for m in fused]
    # the last triple does not map back to its input width (24 -> 6 != 8): left alone
    assert kinds == ["LayerNorm", "FusedMLPGeluTanh", "FusedMLPReLU", "Linear", "GELU", "Linear"]
    assert len(seq) == 10  # not in place by default
    assert torch.equal(fused[1].fc1.weight, seq[1].weight) and torch.equal(fused[2].fc2.bias, seq[6].bias)

    class Blk(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_in, self.act, self.fc_out = nn.Linear(8, 16), nn.GELU(), nn.Linear(16, 8)
    b = fusion_registry.fuse_modules(Blk(), inplace=True)
    assert isinstance(b.fc_in, FusedMLP) and not isinstance(b.fc_in, (FusedMLPGeluTanh, FusedMLPReLU))
    assert not hasattr(b, "act") and not hasattr(b, "fc_out")

    reg = FusionRegistry()
    reg.register_pattern(FusionPattern("pair", [nn.ReLU, nn.ReLU], lambda ms: nn.Identity()))
    assert reg.find_matching_pattern([nn.ReLU(), nn.ReLU()]).name == "pair"
    assert reg.find_matching_pattern([nn.ReLU(), nn.GELU()]) is None
    assert [type(m).__name__ for m in reg.fuse_modules(nn.Sequential(nn.ReLU(), nn.ReLU(), nn.ReLU()))] == ["Identity", "ReLU"]


def test_paged_kv_cache_allocator():
    """BlockManager / PagedKVCache semantics of baseline/inference.py:1045-1303 (host bookkeeping; cache on CPU here)."""
    from mio.baseline.inference import PagedKVCache
    pc = PagedKVCache(num_blocks=6, block_size=4, num_layers=2, num_heads=2, head_dim=8, dtype=torch.float16, device="cpu")
    k, v = pc.get_physical_caches()
    assert tuple(k.shape) == (6, 2, 4, 2, 8) and tuple(v.shape) == tuple(k.shape)
    pc.allocate_blocks_for_sequence(0, 9)           # 3 blocks
    assert len(pc.get_block_table(0)) == 3 and pc.get_sequence_length(0) == 9
    for _ in range(3):
        pc.append_token(0)                           # 10, 11, 12 -> still 3 blocks
    assert len(pc.get_block_table(0)) == 3 and pc.get_sequence_length(0) == 12
    pc.append_token(0)                               # 13 -> 4th block
    assert len(pc.get_block_table(0)) == 4
    pc.append_token(7)                               # new sequence via append
    assert pc.get_sequence_length(7) == 1 and len(pc.get_block_table(7)) == 1
    assert len(set(pc.get_block_table(0)) | set(pc.get_block_table(7))) == 5
    bt, cl, mx = pc.kernel_metadata([0, 7])
    assert bt.dtype == torch.int32 and tuple(bt.shape) == (2, 4) and cl.tolist() == [13, 1] and mx == 13
    assert bt[0].tolist() == pc.get_block_table(0) and bt[1, 0].item() == pc.get_block_table(7)[0]
    with pytest.raises(MemoryError):                 # 1 block left, 3 needed: raises and frees the sequence
        pc.allocate_blocks_for_sequence(9, 12)
    assert pc.get_sequence_length(9) == 0
    with pytest.raises(ValueError):
        pc.get_block_table(9)
    bm = pc.block_manager
    blk = pc.get_block_table(7)[0]
    bm.increase_ref_count(blk)
    pc.free_sequence(7)
    assert blk not in bm.free_blocks                 # still referenced once
    bm.free_block(blk)
    assert blk in bm.free_blocks
    pc.free_sequence(0)
    u = pc.get_memory_usage()
    assert u["free_physical_blocks"] == 6 and u["active_sequences"] == 0 and u["gpu_cache_k_shape"] == (6, 2, 4, 2, 8)


def test_ring_attention_config_and_shells():
    """RingAttentionConfig validation (ring_attention.py:65-89) and the parameter names of the module shells."""
    from mio.kernels.attention import RingAttentionConfig, RingCrossAttention, RingSelfAttention
    for bad in (dict(world_size=0), dict(chunk_size=0), dict(precision="fp8"), dict(attention_dropout=1.0)):
        with pytest.raises(ValueError):
            RingAttentionConfig(**bad)
    assert RingAttentionConfig(precision="fp16").compute_dtype == torch.float16
    with pytest.raises(ValueError):
        RingSelfAttention(30, 4, RingAttentionConfig())
    fused = RingSelfAttention(32, 4, RingAttentionConfig(fuse_qkv=True))
    split = RingSelfAttention(32, 4, RingAttentionConfig(fuse_qkv=False))
    cross = RingCrossAttention(32, 4, RingAttentionConfig())
    assert {n for n, _ in fused.named_parameters()} == {"qkv_proj.weight", "qkv_proj.bias", "out_proj.weight", "out_proj.bias"}
    names = {"q_proj.weight", "q_proj.bias", "k_proj.weight", "k_proj.bias", "v_proj.weight", "v_proj.bias",
             "out_proj.weight", "out_proj.bias"}
    assert {n for n, _ in split.named_parameters()} == names == {n for n, _ in cross.named_parameters()}
    with pytest.raises(ValueError):  # CPU tensor: no fallback
        cross(torch.zeros(1, 4, 32), torch.zeros(1, 4, 32))


@pytest.mark.parametrize("type_id,D", [(0, 64), (1, 64), (0, 96), (0, 128)])
def test_fwd3_accumulator_registers_untouched_by_compiler(tmp_path, type_id, D):
    """fa3_fwd3_kernel keeps O^T, L, Q and the ones operand in accumulator registers Fa3Map<D>::A_Q .. a255 that only
    its inline asm names.  The allocator does not know they are live between asm statements, so the build is only sound
    if no compiler-generated instruction touches them: check the ISA of every instantiation (tools/check_agpr.py)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "ml-inference-optimizer_amd", "csrc")
    isa = tmp_path / "fa.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I../../include", "-I.", "-Wno-unused-value",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", f"-DFA_TYPE_ID={type_id}", f"-DFA_D={D}", "-S",
                    "--cuda-device-only", "fa3_fwd_inst.hip", "-o", str(isa)], cwd=csrc, check=True, capture_output=True)
    text = isa.read_text().splitlines()
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z15fa3_fwd3_kernel\w+:", l)]
    assert len(starts) >= 2, "causal and full instantiations expected"
    floor = 16 * (14 - 2 * (D // 32)) - 4 - 8 * (D // 16)  # Fa3Map<D>::A_Q
    for a in starts:
        b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
        part = tmp_path / "k.s"
        part.write_text("\n".join(text[a:b + 1]))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_agpr.py"), str(part), str(floor)],
                           capture_output=True, text=True)
        assert r.returncode == 0, text[a] + "\n" + r.stdout
        assert not any("scratch_" in l for l in text[a:b + 1]), "register spills in " + text[a]


def test_fwd4_two_waves_per_simd_fits_without_spills(tmp_path):
    """fa3_fwd4_kernel / fa3_fwd5_kernel run two waves per SIMD: 256 registers per wave.  It must fit them without scratch (a staggered
    variant that did not fit ran 60 % slower) -- check the ISA metadata of both dtypes' causal instantiation."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "ml-inference-optimizer_amd", "csrc")
    for type_id in (0, 1):
        isa = tmp_path / f"fa{type_id}.s"
        subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I../../include", "-I.", "-Wno-unused-value",
                        "-Wno-inline-asm", "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", f"-DFA_TYPE_ID={type_id}",
                        "-DFA_D=64", "-S", "--cuda-device-only", "fa3_fwd_inst.hip", "-o", str(isa)], cwd=csrc, check=True,
                       capture_output=True)
        text = isa.read_text()
        blocks = re.findall(r"\.name:\s+_Z15fa3_fwd[45]_kernel\w+\n(?:.*\n){0,12}", text)
        assert not any("fa3_fwd4" in b for b in blocks), "fa3_fwd4_kernel is diagnostic-only since round 3"
        assert len(blocks) >= 8, "fa3_fwd5_kernel: causal / full x {pre-scaled, plain K, ring carry, blocked output}"
        for blk in blocks:
            assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk), blk
            assert re.search(r"\.vgpr_spill_count:\s+0\b", blk), blk
            assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 256, blk


@pytest.mark.parametrize("D", [64, 96, 128])
def test_fwd2_accumulator_registers_untouched_by_compiler(tmp_path, D):
    """fa3_fwd2_kernel keeps its 2*D/32 O^T tiles in the top accumulator registers a[256 - 32*D/32*... : 255] through
    inline asm only; same soundness condition as for fwd3 (a version with the tiles at a0.. had the compiler park
    temporaries in a0-a2 inside the rescale path at D = 96 / 128)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "ml-inference-optimizer_amd", "csrc")
    isa = tmp_path / "fa.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I../../include", "-I.", "-Wno-unused-value",
                    "-mllvm", "-amdgpu-mfma-vgpr-form", "-fno-slp-vectorize", "-DFA_TYPE_ID=0", f"-DFA_D={D}", "-DMIO_DIAG", "-S",
                    "--cuda-device-only", "fa3_fwd_inst.hip", "-o", str(isa)], cwd=csrc, check=True, capture_output=True)
    text = isa.read_text().splitlines()  # (fa3_fwd2 is instantiated in the diagnostic build only: A/B runs)
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z15fa3_fwd2_kernel\w+:", l)]
    assert len(starts) == 2
    floor = 256 - 16 * (2 * D // 32)
    for a in starts:
        b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
        part = tmp_path / "k.s"
        part.write_text("\n".join(text[a:b + 1]))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_agpr.py"), str(part), str(floor)],
                           capture_output=True, text=True)
        assert r.returncode == 0, text[a] + "\n" + r.stdout


def test_gemm_kernels_isa_soundness(tmp_path):
    """(a) gemm8w_kernel (the product's 256x256-tile GEMM, two waves per SIMD, compiler-managed registers): every shipped
    instantiation fits 256 registers without scratch -- a spill inside its K loop would put a vmcnt(0) in front of the
    counted waits.  (b) gemm4w16_kernel / gemm4w16p_kernel (diagnostic library, A/B runs) own the WHOLE accumulator file
    through inline asm: no compiler-generated instruction may touch any accumulator register, and nothing may spill."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "ml-inference-optimizer_amd", "csrc")
    isa = tmp_path / "gemm.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-I../../include", "-I.", "-Wno-unused-value",
                    "-Wno-inline-asm", "-DGEMM_TYPE_ID=0", "-DMIO_DIAG", "-S", "--cuda-device-only", "gemm_inst.hip", "-o", str(isa)],
                   cwd=csrc, check=True, capture_output=True)
    full = isa.read_text()
    text = full.splitlines()
    blocks = re.findall(r"\.name:\s+_Z13gemm8w_kernel\w+Li0EEv7GemmDev\n(?:.*\n){0,12}", full)  # VAR = 0: what the product launches
    assert len(blocks) >= 9, "gemm8w_kernel: 5 plain + 4 residual instantiations (+ SwiGLU) expected"
    for blk in blocks:
        erf = "DF16bLi2E" in blk  # exact-erf GELU: erff() in the read-out may spill a few registers (slow, still correct)
        if not erf:
            assert re.search(r"\.private_segment_fixed_size:\s+0\b", blk), blk
            assert re.search(r"\.vgpr_spill_count:\s+0\b", blk), blk
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1)) <= 256, blk
    # (c) no store whose data registers are rewritten by the very next instruction (tools/check_store_war.py: the hazard
    # behind round 3's wrong lanes -- hipcc leaves no wait state behind a buffer_store with a register soffset)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_store_war.py"), str(isa), "gemm8w_kernel", "1"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    starts = [i for i, l in enumerate(text) if re.match(r"^_Z1[56]gemm4w16p?_kernel\w+:", l)]
    assert len(starts) >= 10
    for a in starts:
        b = next(i for i in range(a, len(text)) if "s_endpgm" in text[i])
        part = tmp_path / "k.s"
        part.write_text("\n".join(text[a:b + 1]))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_agpr.py"), str(part), "0"],
                           capture_output=True, text=True)
        assert r.returncode == 0, text[a] + "\n" + r.stdout
        production = re.match(r"^_Z16gemm4w16p_kernel\w+Lb0EEv7GemmDev:", text[a]) or \
            re.match(r"^_Z15gemm4w16_kernel\w+ELi0EEv7GemmDev:", text[a])  # not the stamp / ablation builds
        erf = "DF16bLi2E" in text[a]
        if production and not erf:
            assert not any("scratch_" in l for l in text[a:b + 1]), "register spills in " + text[a]


@pytest.mark.parametrize("n1,n2", [(64, 4), (13, 4), (4, 4), (5, 1), (1, 0), (3, 0), (12, 8), (2, 1), (7, 3), (0, 0)])
def test_fwd5_tile_stream_schedule(n1, n2):
    """Host-side model of fa3_fwd5_kernel's LDS schedule (csrc/fa3_fwd5_kernel.h): the KV tiles of the one or two causal passes
    of a workgroup are one stream of virtual tiles in 8 LDS stages; a wave meets the barrier once per two iterations (waves 0-3
    behind the QK^T half of an even iteration, waves 4-7 in front of it) and requests the next two tiles there.  Checked for
    every wave, whatever its own tile count: (a) all waves execute the same number of barriers, (b) every tile a half reads
    was requested BEFORE an earlier barrier of that wave (so its vmcnt(0) + the barrier made it visible), (c) a request never
    overwrites a stage that some wave may still read, (d) no tile past the stream's end is requested."""
    STAGES = 8
    passes = [n for n in (n1, n2) if True][: (2 if n2 > 0 else 1)]
    total = sum(passes)

    def wave_events(late, n_w_per_pass):
        """-> list of events in program order: ("sync",) | ("read", virtual_tile) ; sync = vmcnt(0) + barrier + 2 requests"""
        ev, tbase = [], 0
        for pi, n_tiles in enumerate(passes):
            n_w = min(n_w_per_pass[pi], n_tiles)
            ev.append(("pass_start", tbase, n_tiles))
            if n_w > 0:
                ev.append(("read", tbase + 0))       # K(0): scores of tile 0
                ev.append(("read", tbase + 1))       # K(1) fragments for iteration 0 (garbage if the pass has one tile)
            for t in range(n_w):
                even = (t % 2 == 0)
                if late and even:
                    ev.append(("sync",))
                ev.append(("read", tbase + t))       # V(t) fragments, requested in the QK^T half
                if (not late) and even:
                    ev.append(("sync",))
                ev.append(("read", tbase + t + 2))   # K(t+2) fragments, requested in the PV half
            for t in range(n_w, n_tiles):
                if t % 2 == 0:
                    ev.append(("sync",))
            tbase += n_tiles
        return ev

    # the shared (wave-uniform) request stream: what each sync / pass start requests
    def requests():
        out, vnext, vseen, tbase = [], 0, 0, 0    # out[k] = tiles requested at the k-th "request point"
        pts = []
        for pi, n_tiles in enumerate(passes):
            nxt = passes[pi + 1] if pi + 1 < len(passes) else 0
            first = []
            while vnext < tbase + 4:
                first.append(vnext); vnext += 1
            need_barrier = vseen < tbase + 2
            if need_barrier:
                vseen = vnext
            pts.append(("pass_start", [v for v in first if v - tbase < n_tiles + nxt], need_barrier))
            for t in range(n_tiles):
                if t % 2 == 0:
                    vseen = vnext
                    pts.append(("sync", [v for v in (vnext, vnext + 1) if v - tbase < n_tiles + nxt], True))
                    vnext += 2
            tbase += n_tiles
        return pts

    pts = requests()
    assert all(v < total for _, vs, _ in pts for v in vs), "a tile past the end of the stream is requested"          # (d)
    assert sorted(v for _, vs, _ in pts for v in vs) == list(range(total)), "every tile is requested exactly once"
    n_sync = sum(1 for kind, _, _ in pts if kind == "sync")
    waves = []
    for late in (False, True):
        for short in (0, 1, 2, 3):
            n_w = [max(0, n - short) for n in passes]
            ev = wave_events(late, n_w)
            assert sum(1 for e in ev if e[0] == "sync") == n_sync                                                    # (a)
            waves.append((late, ev))
    # replay: global order = request points in sequence; a wave's reads between its (k)th and (k+1)th barrier-like point
    for late, ev in waves:
        k = -1                      # index of the last request point this wave has passed
        visible = set()             # tiles requested before the last vmcnt(0)+barrier this wave executed
        pending = set()             # requested by this wave since
        pi = -1
        for e in ev:
            if e[0] == "pass_start":
                k += 1
                kind, vs, barrier = pts[k]
                assert kind == "pass_start"
                if barrier:
                    pending |= set(vs); visible |= pending; pending = set()
                else:
                    pending |= set(vs)
            elif e[0] == "sync":
                k += 1
                kind, vs, _ = pts[k]
                assert kind == "sync"
                visible |= pending; pending = set(vs)
            else:
                v = e[1]
                if v < total:                                                                                      # (b)
                    assert v in visible, f"late={late}: tile {v} read before a barrier made it visible"
    # (c) WAR: when tile v is requested at request point k, every wave that could still read stage v % 8's previous
    # occupant (tile v - 8) must have passed a barrier after its last read of it.  Waves are at most one barrier apart, so:
    # the last read of tile v - 8 by ANY wave happens before that wave's barrier number (k - 1) at the latest.
    order = {}
    for late, ev in waves:
        k = -1
        for e in ev:
            if e[0] in ("pass_start", "sync"):
                k += 1
            elif e[1] < total:
                order[e[1]] = max(order.get(e[1], -1), k)   # read of tile happens after request point k of that wave
    for k, (_, vs, _) in enumerate(pts):
        for v in vs:
            if v - STAGES >= 0 and (v - STAGES) in order:
                # a wave issuing request point k has passed barrier k; another wave may still be before ITS point k only if
                # point k is a barrier for it too (then it has arrived); reads recorded "after point j" with j <= k - 1 are done
                assert order[v - STAGES] <= k - 1, f"tile {v} overwrites tile {v - STAGES} that may still be read"
