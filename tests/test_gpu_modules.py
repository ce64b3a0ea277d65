"""GPU parity of the operator surface (modules / converters / Optimizer facade / synthetic stack)."""
import copy

import pytest
import torch
import torch.nn.functional as F

import oracle

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).abs().mean() / b.abs().mean()).item()


@pytest.mark.parametrize("dtype,prec", [(torch.bfloat16, "bf16"), (torch.float16, "fp16")])
@pytest.mark.parametrize("cls", ["layer", "self"])
def test_attention_layers(dtype, prec, cls):
    from mio.kernels.attention import FlashAttentionConfig, FlashAttentionLayer, FlashSelfAttention
    torch.manual_seed(0)
    d, H, Hkv, B, S = 128, 8, 4, 2, 150
    cfg = FlashAttentionConfig(causal=True, precision=prec)
    m = (FlashAttentionLayer if cls == "layer" else FlashSelfAttention)(d, H, cfg, num_kv_heads=Hkv)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.1)
    m = m.to(DEV, dtype).eval()
    x = torch.randn(B, S, d, device=DEV, dtype=dtype)
    y = m(x)
    D = d // H
    xf = x.cpu().double()
    sd = {k: v.cpu().double() for k, v in m.state_dict().items()}
    if cls == "layer":
        q = F.linear(xf, sd["q_proj.weight"], sd["q_proj.bias"])
        k = F.linear(xf, sd["k_proj.weight"], sd["k_proj.bias"])
        v = F.linear(xf, sd["v_proj.weight"], sd["v_proj.bias"])
    else:
        qkv = F.linear(xf, sd["qkv_proj.weight"], sd["qkv_proj.bias"])
        q, k, v = qkv[..., :d], qkv[..., d:d + Hkv * D], qkv[..., d + Hkv * D:]
    ctx = oracle.standard_attention(q.view(B, S, H, D), k.reshape(B, S, Hkv, D), v.reshape(B, S, Hkv, D), causal=True)
    ref = F.linear(ctx.reshape(B, S, d), sd["o_proj.weight"], sd["o_proj.bias"])
    assert _rel(y, ref) < (6e-3 if dtype == torch.bfloat16 else 3e-3)  # 4 chained 16-bit roundings (q/k/v, ctx, out)
    # fp32 activations with 16-bit precision configured: cast in, cast back (flash_attention.py:176-225)
    y32 = m.float()(x.float())
    assert y32.dtype == torch.float32 and _rel(y32, ref) < 6e-3


def test_flash_attention3_module_and_mask():
    from mio.kernels.attention import FlashAttention3, FlashAttentionConfig, ModelConverter
    torch.manual_seed(1)
    q, k, v = (torch.randn(2, 100, 4, 64, device=DEV) for _ in range(3))  # fp32 in, fp16 compute (default precision)
    fa = FlashAttention3(FlashAttentionConfig(causal=False))
    mask = torch.ones(2, 100, device=DEV)
    mask[1, 60:] = 0
    o = fa(q, k, v, ModelConverter.convert_mask(mask))
    assert o.dtype == torch.float32
    ref = oracle.standard_attention(q.cpu().half(), k.cpu().half(), v.cpu().half(), mask=mask.cpu())
    assert (o.cpu().double() - ref).abs().max() < 4e-3
    with pytest.raises(NotImplementedError):
        FlashAttention3(FlashAttentionConfig(return_softmax=True))(q, k, v)


def test_paged_path_through_module():
    from mio.kernels.attention import FlashAttentionConfig, FlashSelfAttention
    torch.manual_seed(2)
    d, H, B, bs, L, nblk = 128, 4, 2, 16, 2, 16
    D = d // H
    dt = torch.float16
    m = FlashSelfAttention(d, H, FlashAttentionConfig(precision="fp16")).to(DEV, dt).eval()
    kc = torch.randn(nblk, L, bs, H, D, device=DEV, dtype=dt)
    vc = torch.randn(nblk, L, bs, H, D, device=DEV, dtype=dt)
    bt = torch.stack([torch.randperm(nblk)[:6] for _ in range(B)]).to(torch.int32).to(DEV)
    cl = torch.tensor([70, 33], dtype=torch.int32, device=DEV)
    x = torch.randn(B, 1, d, device=DEV, dtype=dt)
    y = m(x, physical_kv_cache_k=kc, physical_kv_cache_v=vc, block_tables=bt, context_lengths=cl,
          kv_cache_block_size=bs, max_seq_len=96, layer_idx=1)
    sd = {k_: v_.cpu().double() for k_, v_ in m.state_dict().items()}
    q = F.linear(x.cpu().double(), sd["qkv_proj.weight"][:d], sd["qkv_proj.bias"][:d]).view(B, 1, H, D).permute(0, 2, 1, 3)
    ctx = oracle.paged_attention_forward(q, kc.cpu(), vc.cpu(), bt.cpu(), cl.cpu(), bs, 1)
    ref = F.linear(ctx.permute(0, 2, 1, 3).reshape(B, 1, d), sd["o_proj.weight"], sd["o_proj.bias"])
    assert _rel(y, ref) < 5e-3
    with pytest.raises(ValueError):
        m(x, block_tables=bt)


def test_optimizer_facade_end_to_end():
    """Optimizer(model).optimize(...) on the plain GPT-2-shaped stack == the plain model (the reference's
    stated criterion, test_parallelism.py:306-322, at bf16 tolerance instead of 0.1)."""
    from ml_inference_optimizer import Optimizer
    from oracle.baseline_runner import PlainGPT2Stack
    torch.manual_seed(0)
    plain = PlainGPT2Stack(128, 4, 2, seed=5).eval()
    x = torch.randn(2, 96, 128)
    with torch.no_grad():
        ref = plain(x)
    model = copy.deepcopy(plain).to(DEV, torch.bfloat16)
    opt = Optimizer(model).optimize(use_flash_attention=True, use_fused_mlp=True, tensor_parallel_size=1,
                                    use_custom_layernorm=True, causal=True)
    with torch.no_grad():
        y = opt(x.to(DEV, torch.bfloat16))
    assert _rel(y, ref) < 1e-2 and (y.float().cpu() - ref).abs().max() < 0.15  # fp32 weights vs their bf16 copies: weight rounding included


def test_synthetic_stack_matches_plain_stack():
    """The benchmark model (mio.synthetic) == the oracle's plain fp32 stack with the same weights."""
    from mio.synthetic import GPT2ShapedStack
    from oracle.baseline_runner import PlainGPT2Stack
    d, H, L, B, S = 128, 4, 3, 2, 200
    plain = PlainGPT2Stack(d, H, L, seed=7).eval()
    fast = GPT2ShapedStack(d, H, L, causal=True, precision="bf16")
    with torch.no_grad():
        for i in range(L):
            a, b = plain.h[i], fast.h[i]
            b.attn.qkv_proj.weight.copy_(torch.cat([a.attn.q_proj.weight, a.attn.k_proj.weight, a.attn.v_proj.weight]))
            b.attn.qkv_proj.bias.copy_(torch.cat([a.attn.q_proj.bias, a.attn.k_proj.bias, a.attn.v_proj.bias]))
            b.attn.o_proj.load_state_dict(a.attn.o_proj.state_dict())
            b.mlp.mlp.fc1.load_state_dict(a.mlp.linear1.state_dict())
            b.mlp.mlp.fc2.load_state_dict(a.mlp.linear2.state_dict())
            b.ln_1.load_state_dict(a.ln_1.state_dict())
            b.ln_2.load_state_dict(a.ln_2.state_dict())
        fast.ln_f.load_state_dict(plain.ln_f.state_dict())
    x = torch.randn(B, S, d)
    with torch.no_grad():
        ref = plain(x)
        y = fast.to(DEV, torch.bfloat16)(x.to(DEV, torch.bfloat16))
    assert _rel(y, ref) < 1e-2  # fp32 weights vs their bf16 copies through 3 layers: weight rounding included


def test_mlp_converter_module_on_gpu():
    from mio.kernels.mlp import FusedTransformerMLP
    torch.manual_seed(3)
    for act, oact in (("gelu", "gelu"), ("swiglu", "swiglu"), ("relu", "relu"), ("silu", "silu")):
        m = FusedTransformerMLP(128, 256, act).to(DEV, torch.float16).eval()
        x = torch.randn(2, 50, 128, device=DEV, dtype=torch.float16)
        y = m(x)
        sd = {k: v.cpu() for k, v in m.state_dict().items()}
        ref = oracle.fused_mlp(x.cpu(), sd["mlp.fc1.weight"], sd["mlp.fc1.bias"], sd["mlp.fc2.weight"], sd["mlp.fc2.bias"],
                               oact, sd.get("mlp.fc1_gate.weight"), sd.get("mlp.fc1_gate.bias"))
        assert _rel(y, ref) < 3e-3


@pytest.mark.parametrize("fuse_qkv", [True, False])
def test_ring_attention_modules(fuse_qkv):
    """RingSelfAttention / RingCrossAttention shells (ring_attention.py:168-669) against the oracle's exact
    attention on the same projections; additive mask [B,1,Sq,Sk]."""
    from mio.kernels.attention import RingAttentionConfig, RingCrossAttention, RingSelfAttention
    torch.manual_seed(3)
    d, H, B, Sq, Sk = 128, 2, 2, 140, 200
    dtype = torch.float16
    cfg = RingAttentionConfig(fuse_qkv=fuse_qkv, precision="fp16")
    x = torch.randn(B, Sq, d, dtype=dtype)
    ctx = torch.randn(B, Sk, d, dtype=dtype)

    def heads(t):
        return t.view(t.shape[0], t.shape[1], H, d // H)

    m = RingSelfAttention(d, H, cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
    m = m.to(DEV, dtype).eval()
    sd = {k: v.cpu().double() for k, v in m.state_dict().items()}
    mask = torch.zeros(B, 1, Sq, Sq)
    mask[:, :, :, Sq - 17:] = -1e9
    y = m(x.to(DEV), attention_mask=mask.to(DEV))
    xf = x.double()
    if fuse_qkv:
        qkv = F.linear(xf, sd["qkv_proj.weight"], sd["qkv_proj.bias"]).view(B, Sq, 3, H, d // H)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]
    else:
        q, k, v = (heads(F.linear(xf, sd[f"{n}_proj.weight"], sd[f"{n}_proj.bias"])) for n in "qkv")
    ref = oracle.standard_attention(q, k, v, additive_mask=mask.double()).reshape(B, Sq, d)
    ref = F.linear(ref, sd["out_proj.weight"], sd["out_proj.bias"])
    assert _rel(y, ref) < 3e-3

    c = RingCrossAttention(d, H, cfg)
    with torch.no_grad():
        for p in c.parameters():
            p.copy_(torch.randn_like(p) * 0.08)
    c = c.to(DEV, dtype).eval()
    sd = {k: v.cpu().double() for k, v in c.state_dict().items()}
    y = c(x.to(DEV), ctx.to(DEV))
    q = heads(F.linear(x.double(), sd["q_proj.weight"], sd["q_proj.bias"]))
    k = heads(F.linear(ctx.double(), sd["k_proj.weight"], sd["k_proj.bias"]))
    v = heads(F.linear(ctx.double(), sd["v_proj.weight"], sd["v_proj.bias"]))
    ref = F.linear(oracle.standard_attention(q, k, v).reshape(B, Sq, d), sd["out_proj.weight"], sd["out_proj.bias"])
    assert _rel(y, ref) < 3e-3


def test_fusion_registry_on_gpu():
    """fusion_registry.fuse_modules output == the unfused torch modules (exact and tanh GELU, ReLU)."""
    from mio.baseline.inference import fusion_registry
    torch.manual_seed(4)
    dtype = torch.float16
    seq = torch.nn.Sequential(torch.nn.Linear(96, 256), torch.nn.GELU(), torch.nn.Linear(256, 96),
                              torch.nn.Linear(96, 160), torch.nn.GELU(approximate="tanh"), torch.nn.Linear(160, 96),
                              torch.nn.Linear(96, 128), torch.nn.ReLU(), torch.nn.Linear(128, 96)).to(DEV, dtype).eval()
    fused = fusion_registry.fuse_modules(seq).eval()
    assert [type(m).__name__ for m in fused] == ["FusedMLP", "FusedMLPGeluTanh", "FusedMLPReLU"]
    x = torch.randn(3, 77, 96, device=DEV, dtype=dtype)
    with torch.no_grad():
        ref = copy.deepcopy(seq).double()(x.double())
        y = fused(x)
    assert _rel(y, ref) < 3e-3


def test_paged_kv_cache_decode_loop():
    """PagedKVCache (inference.py:1150-1303) driving reshape_and_cache + paged decode for ragged sequences, token by
    token, against dense attention over everything written so far."""
    from mio import ops
    from mio.baseline.inference import PagedKVCache
    torch.manual_seed(5)
    H, D, L, bs, dtype = 4, 64, 2, 16, torch.float16
    pc = PagedKVCache(num_blocks=24, block_size=bs, num_layers=L, num_heads=H, head_dim=D, dtype=dtype, device=DEV)
    k_cache, v_cache = pc.get_physical_caches()
    lens0 = [37, 5, 16]
    hist = {s: ([], []) for s in range(len(lens0))}
    layer = 1
    for step in range(max(lens0) + 3):
        active = [s for s, n in enumerate(lens0) if step < n + 3]
        for s in active:
            pc.append_token(s)
        bt, cl, mx = pc.kernel_metadata(active)
        kk = torch.randn(len(active), 1, H, D, dtype=dtype)   # [B, 1, Hkv, D] (attention_kernels.py:1314-1323)
        vv = torch.randn(len(active), 1, H, D, dtype=dtype)
        for i, s in enumerate(active):
            hist[s][0].append(kk[i, 0])
            hist[s][1].append(vv[i, 0])
        ops.reshape_and_cache(kk.to(DEV), vv.to(DEV), k_cache, v_cache, bt, cl, bs, layer)
        if step % 7 == 0 or step >= max(lens0):
            q = torch.randn(len(active), H, 1, D, dtype=dtype)
            o = torch.empty(len(active), H, 1, D, dtype=dtype, device=DEV)
            ops.paged_attention_forward(q.to(DEV), o, k_cache, v_cache, bt, cl, bs, mx, layer)
            for i, s in enumerate(active):
                K = torch.stack(hist[s][0], 0)[None].double()   # [1, S, H, D]
                V = torch.stack(hist[s][1], 0)[None].double()
                ref = oracle.standard_attention(q[i].permute(1, 0, 2)[None].double(), K, V)[0].permute(1, 0, 2)
                assert _rel(o[i], ref) < 2e-3, (step, s)
    assert (k_cache[:, 0] == 0).all()  # the other layer of the cache was never touched
    for s in range(len(lens0)):
        pc.free_sequence(s)
    assert pc.get_memory_usage()["free_physical_blocks"] == 24


def test_composed_functional_entry_points():
    """fused_attention (flash_attention_kernels.py:1361-1530) and fused LayerNorm + QKV with its two adapters
    (fused_layernorm_qkv.py:422-700, 1073-1161) against the same math in fp64 on the CPU."""
    from mio import ops
    torch.manual_seed(6)
    B, S, d, H, dtype = 2, 130, 128, 2, torch.float16
    x = torch.randn(B, S, d, dtype=dtype)
    wqkv, bqkv = (torch.randn(3 * d, d) * 0.06).to(dtype), (torch.randn(3 * d) * 0.1).to(dtype)
    wo, bo = (torch.randn(d, d) * 0.06).to(dtype), (torch.randn(d) * 0.1).to(dtype)
    g, b = (1 + 0.1 * torch.randn(d)).to(dtype), (0.1 * torch.randn(d)).to(dtype)
    dev = lambda *ts: [t.to(DEV) for t in ts]
    y = ops.triton_fused_attention(*dev(x, wqkv, bqkv, wo, bo), causal=True, num_heads=H)
    qkv = F.linear(x.double(), wqkv.double(), bqkv.double()).view(B, S, 3, H, d // H)
    ref = oracle.standard_attention(qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2], causal=True).reshape(B, S, d)
    ref = F.linear(ref, wo.double(), bo.double())
    assert _rel(y, ref) < 3e-3

    xn = F.layer_norm(x.double(), (d,), g.double(), b.double(), 1e-5)
    qr, kr, vr = (F.linear(xn, wqkv.double()[i * d:(i + 1) * d], bqkv.double()[i * d:(i + 1) * d]) for i in range(3))
    q, k, v = ops.flash_compatible_wrapper(*dev(x, g, b, wqkv, bqkv), num_heads=H)
    assert tuple(q.shape) == (B, S, H, d // H)
    for got, want in ((q, qr), (k, kr), (v, vr)):
        assert _rel(got.reshape(B, S, d), want) < 3e-3
    ws = [wqkv[i * d:(i + 1) * d].contiguous() for i in range(3)]
    bs = [bqkv[i * d:(i + 1) * d].contiguous() for i in range(3)]
    q2, k2, v2 = ops.triton_fused_layernorm_qkv(*dev(x, g, b, *ws, *bs), num_heads=H)
    assert torch.equal(q2, q) and torch.equal(k2, k) and torch.equal(v2, v)
    q3, k3, v3 = ops.ring_compatible_wrapper(*dev(x, g, b, *ws, *bs), num_heads=H)
    assert tuple(q3.shape) == (B, H, S, d // H) and torch.equal(q3.permute(0, 2, 1, 3), q)
    assert ops._infer_heads(1280, 0) == 20 and ops._infer_heads(1024, 0) == 16 and ops._infer_heads(96, 0) == 1


@pytest.mark.parametrize("d,H", [(2048, 16), (1280, 16)])
def test_block_other_head_dims_layernorm_folded(d, H):
    """Head dims the pre-scaled-K / blocked-output attention kernels do not take (128: LLaMA-class; 80: blocked output needs <= 64):
    the stream form still folds both LayerNorms -- QKV normalises in its read-out, the attention kernel for that head dim runs on
    plain / pre-scaled K, the output projection takes its row-major (or blocked) context and writes stream + statistics."""
    from mio.synthetic import Block
    from mio._nn import ResidualStream
    torch.manual_seed(10)
    I, B, S = 2 * d, (2 if d == 2048 else 4), 4096   # (>= 256 output tiles on the N = d GEMMs)
    blk = Block(d, H, I, causal=True, precision="bf16").to(DEV, torch.bfloat16).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn_like(p_) * 0.02)
        blk.ln_1.weight.add_(1.0)
        blk.ln_2.weight.add_(1.0)
        x = torch.randn(B, S, d, device=DEV, dtype=torch.bfloat16) + 0.3
        assert blk.stream_ok(B, S, torch.bfloat16)
        s1 = blk(x, stream_out=True)
        assert isinstance(s1, ResidualStream)
        y = blk(s1)                                   # a second pass through the same block: ln_1 folded as well
        ref1 = blk(x, fold=False)
        ref = blk(ref1, fold=False)
    rel1 = ((s1.dense().float() - ref1.float()).abs().mean() / ref1.float().abs().mean()).item()
    rel = ((y.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    assert rel1 < 3e-3 and rel < 4e-3, (rel1, rel)


def test_block_swiglu_layernorm_folded():
    """synthetic.Block with the SwiGLU MLP at a size where the LayerNorms fold into the GEMMs (ResidualStream): ln_2 runs inside
    the gated stage's read-out (interleaved gate / up weight, both halves scaled by rstd); against the same block with separate
    LayerNorm kernels."""
    from mio.synthetic import Block
    torch.manual_seed(9)
    d, H, I, B, S = 1024, 16, 2048, 4, 4096
    blk = Block(d, H, I, causal=True, precision="bf16", activation="swiglu").to(DEV, torch.bfloat16).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn_like(p_) * 0.03)
        blk.ln_1.weight.add_(1.0)
        blk.ln_2.weight.add_(1.0)
        x = torch.randn(B, S, d, device=DEV, dtype=torch.bfloat16) + 0.3
        assert blk.stream_ok(B, S, torch.bfloat16)
        y = blk(x)
        ref = blk(x, fold=False)
    rel = ((y.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    assert rel < 3e-3, rel


@pytest.mark.parametrize("precision,dtype", [("fp16", torch.float16), ("bf16", torch.bfloat16)])
def test_stack_layernorm_fold_fp16_and_ragged_rows(precision, dtype):
    """GPT2ShapedStack (3 blocks) with the LayerNorms folded, fp16 and bf16, at a token count that is not a multiple of the 256-row
    tiles (B 5 x S 3301 = 16 505 rows: the blocked stream and its statistics are padded, the padding rows are never stored): against
    the same stack with separate LayerNorm kernels; and the fold is what ran (two LayerNorm launches' worth of stream objects)."""
    from mio.synthetic import GPT2ShapedStack
    from mio._nn import ResidualStream
    torch.manual_seed(12)
    d, H, L, I, B, S = 1024, 16, 3, 2048, 5, 3301
    stack = GPT2ShapedStack(d, H, L, I, causal=True, precision=precision, seed=4).to(DEV, dtype).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(6)
        for m in stack.modules():
            if isinstance(m, torch.nn.Linear):
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.02)
            if isinstance(m, torch.nn.LayerNorm):
                m.weight.copy_(1 + 0.1 * torch.randn(m.weight.shape, generator=g))
                m.bias.copy_(0.1 * torch.randn(m.bias.shape, generator=g))
    x = (torch.randn(B, S, d, generator=torch.Generator().manual_seed(7)) + 0.25).to(DEV, dtype)
    assert all(blk.stream_ok(B, S, dtype) for blk in stack.h)
    assert isinstance(stack.h[0](x, stream_out=True), ResidualStream)
    y = stack(x)
    stack.no_ln_fold = True
    ref = stack(x)
    assert torch.isfinite(y).all()
    rel = ((y.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    assert rel < (2e-3 if dtype == torch.float16 else 6e-3), rel


def test_block_prenorm_equals_decomposed():
    """synthetic.Block at a size where LayerNorm hands over in the blocked layout (pre_norm=) == the same block with
    the LayerNorms applied outside the modules (plain layout): same kernels, same arithmetic -> same bits."""
    from mio.synthetic import Block
    torch.manual_seed(8)
    d, H, I, B, S = 1024, 16, 1024, 4, 4096 + 25
    blk = Block(d, H, I, causal=True, precision="bf16").to(DEV, torch.bfloat16).eval()
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.copy_(torch.randn_like(p_) * 0.03)
        blk.ln_1.weight.add_(1.0)
        blk.ln_2.weight.add_(1.0)
        x = torch.randn(B, S, d, device=DEV, dtype=torch.bfloat16)
        y = blk(x, fold=False)
        a = blk.attn(blk.ln_1(x), residual=x)
        ref = blk.mlp(blk.ln_2(a), residual=a)
        y_fold = blk(x)   # ln_2 folded into the GEMMs around it (ResidualStream): other roundings, same values
    assert torch.equal(y, ref)
    assert blk.stream_ok(B, S, torch.bfloat16)
    rel = ((y_fold.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item()
    assert rel < 3e-3, rel
