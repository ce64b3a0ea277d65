#!/usr/bin/env python3
"""What the multi-GPU lines of bench.py SHOULD show on one 8 x MI355X node, from the single-GPU kernel times and the xGMI link
rate -- written down before any such run exists, so that a SCALE / extra.* result can be checked against it (DESIGN.md
section 5).  No hardware needed: `python tools/scale_model.py` prints the table; bench.py attaches the same numbers as
extra.*.predicted.

Model (all per GPT-2-shaped layer, d 1024, H 16, I 4096, bf16, B 8 x S 4096 tokens per replica / tensor group):
  GEMM       t(M, N, K) = ceil(tiles / 256) * (K / 32 * T_K + T_TILE [+ T_RES]) + T_RAMP,  tiles = ceil(M/256) * ceil(N/256)
             (gemm8w_kernel.h: 0.69 us per K-tile of 32 on random data, 6.5 us per tile of read-out / stores / boundary,
             +2.5 us with the residual operand, 8 us of launch ramp and last-tile flush; reproduces the four measured C2 GEMMs
             within 7 %)
  attention  FLOPs / 993 TFLOP/s (fa3_fwd5, causal, measured at B 8 H 16 S 4096), heads / tp per tensor-parallel rank;
             one ring step B 1 H 16 8192 x 8192 = 0.2455 ms measured (tools/dbg/ring_step.py)
  LayerNorm  0.0217 ms each (replicated under tensor parallelism)
  boundary   0.055 ms per layer of inter-kernel gaps (measured step minus the sum of its kernels)
  links      one xGMI link = 153 GB/s per direction nominal, 7 per GPU (full mesh); LINK_EFF of it reached by RCCL (0.8 assumed)
  all-reduce of S bytes on t ranks: ring 2 (t-1)/t S over ONE link; mesh (reduce-scatter + all-gather over the t-1 direct
             links at once) 2 S / t per link
"""
import json
import math

T_K, T_TILE, T_RES, T_RAMP = 0.69e-3, 6.5e-3, 2.5e-3, 8e-3  # ms
ATTN_TFLOPS = 993.0
T_LN = 0.0217
T_GAP_LAYER = 0.055
T_RING_STEP = 0.2455       # ms, B 1 H 16 8192 x 8192 non-causal, k_prescaled carry
LINK_GBS, LINK_EFF = 153.0, 0.8


def gemm_ms(M, N, K, res=False, act_extra=0.0):
    tiles = math.ceil(M / 256) * math.ceil(N / 256)
    return math.ceil(tiles / 256) * (K / 32 * T_K + T_TILE + (T_RES if res else 0.0) + act_extra) + T_RAMP


def attn_ms(B, S, H, D, causal=True):
    fl = (2.0 * B * S * (S + 1) if causal else 4.0 * B * S * S) * H * D
    return fl / (ATTN_TFLOPS * 1e12) * 1e3


def layer_ms(B=8, S=4096, d=1024, H=16, I=4096, tp=1):
    M = B * S
    g = dict(qkv=gemm_ms(M, 3 * d // tp, d), oproj=gemm_ms(M, d, d // tp, res=True),
             fc1=gemm_ms(M, I // tp, d, act_extra=1.0e-3), fc2=gemm_ms(M, d, I // tp, res=True))
    a = attn_ms(B, S, H // tp, d // H)
    return g, a, sum(g.values()) + a + 2 * T_LN + T_GAP_LAYER


def allreduce_ms(nbytes, t, how):
    bw = LINK_GBS * 1e9 * LINK_EFF
    per_link = 2.0 * (t - 1) / t * nbytes if how == "ring" else 2.0 * nbytes / t
    return per_link / bw * 1e3


def predict(L=24, B=8, S=4096, d=1024, H=16, I=4096, ring_S=65536, measured_single_ms=None):
    out = {}
    _, _, l1 = layer_ms(B, S, d, H, I, 1)
    single = measured_single_ms if measured_single_ms else L * l1
    out["single_gpu"] = {"ms_per_step": round(single, 2), "tokens_per_s": round(B * S / single * 1e3), "model_ms_per_step": round(L * l1, 2)}
    for n in (2, 4, 8):  # data parallel: replicas only, no data-path collective
        out[f"dp{n}"] = {"ms_per_step": round(single, 2), "tokens_per_s": round(n * B * S / single * 1e3), "efficiency": 1.0}
    payload = B * S * d * 2
    for tp in (2, 4):
        g, a, lt = layer_ms(B, S, d, H, I, tp)
        row = {"compute_ms_per_layer": round(lt, 4), "gemm_ms": {k: round(v, 4) for k, v in g.items()}, "attention_ms": round(a, 4)}
        for how in ("ring", "mesh"):
            ar = allreduce_ms(payload, tp, how)
            # unoverlapped: two blocking all-reduces per layer; overlapped: chunk i's all-reduce under chunk i + 1's GEMM
            # (4 chunks): what sticks out is max(0, AR - GEMM) + one chunk of the slower of the two
            un = lt + 2 * ar
            ov = lt + sum(max(0.0, ar - t) + min(ar, t) / 4 for t in (g["oproj"], g["fc2"]))
            row[how] = {"allreduce_ms": round(ar, 4), "unoverlapped_ms_per_step": round(L * un, 2), "overlapped_ms_per_step": round(L * ov, 2),
                        "unoverlapped_tokens_per_s_per_group": round(B * S / (L * un) * 1e3),
                        "overlapped_tokens_per_s_per_group": round(B * S / (L * ov) * 1e3)}
        out[f"tensor_parallel_tp{tp}"] = row
    # ring attention core, B 1, S 65536 over 8 ranks (attention only, q / k / v resident: what bench_ring times)
    sp = 8
    shard = 2 * (ring_S // sp) * d * 2  # K + V shard bytes
    x1 = shard / (LINK_GBS * 1e9 * LINK_EFF) * 1e3
    comp = sp * T_RING_STEP
    zz = (sp * 2 * (sp * 2 + 1) / 2 / sp) / 4.0  # half-block pairs per rank / 4 = equivalent full steps (diagonals count half: -1/4)
    zz_ms = (zz - 0.25) * T_RING_STEP * 1.1
    out[f"ring_attention_sp{sp}"] = {
        "exchange_ms_per_shard_per_link": round(x1, 4), "compute_ms_noncausal": round(comp, 3),
        "noncausal_ring": {"ms": round(comp + (sp - 1) * max(0.0, x1 - T_RING_STEP), 3)},
        "noncausal_ring_unoverlapped": {"ms": round(comp + (sp - 1) * x1, 3)},
        "noncausal_mesh": {"ms": round(comp + max(0.0, x1 - T_RING_STEP), 3)},
        "noncausal_mesh_unoverlapped": {"ms": round(comp + x1, 3)},
        "causal_zigzag_mesh": {"ms": round(zz_ms + max(0.0, x1 - zz_ms / sp), 3)},
    }
    for k, v in out[f"ring_attention_sp{sp}"].items():
        if isinstance(v, dict):
            v["tokens_per_s"] = round(ring_S / v["ms"] * 1e3)
    return out


if __name__ == "__main__":
    print(json.dumps(predict(), indent=1))
