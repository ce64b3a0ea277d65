#!/usr/bin/env python3
"""Board power, gfx clock and ENERGY PER LAUNCH of the benchmark's kernels and of the whole forward (amdsmi samples
beside the running work): the direct evidence for DESIGN.md section 4.2's "the step is power-bound".

    python tools/power_trace.py [--seconds 2.0] [--out gpurun_out/power_trace.json]

For each workload (idle, the C2 forward, and each of its kernels queued back to back) a sampler thread reads
amdsmi's socket power, per-XCD gfx clocks, the energy accumulator and the violation (throttle) residencies every
~5 ms while the stream stays full.  Reported: average / max power, average clock, J per launch (energy counter delta /
launches), ms per launch, and the PPT (power) / thermal violation percentages amdsmi reports.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


class Sampler:
    def __init__(self, period=0.005):
        import amdsmi

        self.smi = amdsmi
        amdsmi.amdsmi_init()
        self.handles = amdsmi.amdsmi_get_processor_handles()
        self.period = period
        self.h = self.handles[0]
        self.samples = []
        self.stop = False
        self.thread = None

    def pick(self, fn):
        """More than one device visible to amdsmi: pick the one whose power rises while fn() keeps cuda:0 busy."""
        if len(self.handles) == 1:
            return
        base = [self._power(h) for h in self.handles]
        t = threading.Thread(target=fn)
        t.start()
        time.sleep(0.5)
        busy = [self._power(h) for h in self.handles]
        t.join()
        d = [b - a for a, b in zip(base, busy)]
        self.h = self.handles[d.index(max(d))]

    def _power(self, h):
        try:
            p = self.smi.amdsmi_get_power_info(h)
            for k in ("current_socket_power", "socket_power", "average_socket_power"):
                v = p.get(k)
                if isinstance(v, (int, float)) and v > 0:
                    return float(v)
        except Exception:
            pass
        return 0.0

    def _one(self):
        s = {"t": time.perf_counter(), "w": self._power(self.h)}
        try:
            m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
            clk = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and 0 < c < 10000]
            if clk:
                s["mhz"] = sum(clk) / len(clk)
            elif isinstance(m.get("current_gfxclk"), (int, float)):
                s["mhz"] = float(m["current_gfxclk"])
            for k in ("average_gfx_activity", "temperature_hotspot", "current_socket_power", "energy_accumulator"):
                if isinstance(m.get(k), (int, float)):
                    s[k] = m[k]
        except Exception:
            pass
        return s

    def energy(self):
        """Energy accumulator in joules (amdsmi_get_energy_count: counter * resolution in micro-joules)."""
        try:
            e = self.smi.amdsmi_get_energy_count(self.h)
            cnt = e.get("energy_accumulator", e.get("power"))
            res = e.get("counter_resolution", 15.3)
            return float(cnt) * float(res) * 1e-6
        except Exception:
            return None

    def violations(self):
        try:
            v = self.smi.amdsmi_get_violation_status(self.h)
            return {k: v[k] for k in v if isinstance(v[k], (int, float)) and ("per_" in k or "active" in k)}
        except Exception as ex:
            return {"error": str(ex)[:80]}

    def _loop(self):
        while not self.stop:
            self.samples.append(self._one())
            time.sleep(self.period)

    def start(self):
        self.samples, self.stop = [], False
        self.thread = threading.Thread(target=self._loop, daemon=True)
        self.thread.start()

    def finish(self):
        self.stop = True
        self.thread.join()
        return self.samples


def measure(sm, name, fn, seconds, launches_per_call=1):
    import torch

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); fn(); e.record(); torch.cuda.synchronize()
    per = max(s.elapsed_time(e), 1e-3)
    n = max(3, int(seconds * 1e3 / per))
    for _ in range(max(1, int(300.0 / per))):  # reach the sustained operating point first
        fn()
    torch.cuda.synchronize()
    e0 = sm.energy()
    sm.start()
    t0 = time.perf_counter()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    smp = sm.finish()
    e1 = sm.energy()
    viol = sm.violations()
    ms = s.elapsed_time(e) / n
    w = [x["w"] for x in smp if x.get("w")]
    mhz = [x["mhz"] for x in smp if x.get("mhz")]
    out = {"workload": name, "calls": n, "ms_per_call": ms, "samples": len(smp),
           "power_w_avg": sum(w) / len(w) if w else None, "power_w_max": max(w) if w else None,
           "gfx_mhz_avg": sum(mhz) / len(mhz) if mhz else None, "gfx_mhz_min": min(mhz) if mhz else None,
           "violations": viol}
    if e0 is not None and e1 is not None and e1 > e0:
        out["joule_per_call_counter"] = (e1 - e0) / n
        out["watt_from_counter"] = (e1 - e0) / (t1 - t0)
    if w:
        out["joule_per_call_from_power"] = (sum(w) / len(w)) * ms * 1e-3
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "power_trace.json"))
    a = ap.parse_args()
    import torch
    from mio import ops
    from mio.synthetic import GPT2ShapedStack

    dt = torch.bfloat16
    B, S, d, H, L = 8, 4096, 1024, 16, 24
    D, M, I = d // H, B * S, 4 * d
    model = GPT2ShapedStack(d, H, L, I, causal=True, precision="bf16", seed=0).to(device="cuda", dtype=dt).eval()
    torch.manual_seed(0)
    x = torch.randn(B, S, d, device="cuda", dtype=dt)
    sm = Sampler()

    def fwd():
        with torch.no_grad():
            model(x)

    sm.pick(lambda: [fwd() for _ in range(20)] and torch.cuda.synchronize())
    res = {"device": torch.cuda.get_device_name(0), "amdsmi_devices": len(sm.handles)}
    try:
        res["power_cap"] = {k: v for k, v in sm.smi.amdsmi_get_power_cap_info(sm.h).items() if isinstance(v, (int, float))}
    except Exception as ex:
        res["power_cap"] = {"error": str(ex)[:80]}
    time.sleep(1.0)
    sm.start(); time.sleep(1.0); idle = sm.finish()
    w = [s_["w"] for s_ in idle if s_.get("w")]
    res["idle_w"] = sum(w) / len(w) if w else None
    rows = []
    rows.append(measure(sm, "C2 forward (24 layers)", fwd, max(a.seconds, 3.0)))
    blk = model.h[0]
    wqkv, bqkv = blk.attn.qkv_proj.weight, blk.attn.qkv_proj.bias
    wo, bo = blk.attn.o_proj.weight, blk.attn.o_proj.bias
    w1, b1 = blk.mlp.mlp.fc1.weight, blk.mlp.mlp.fc1.bias
    w2, b2 = blk.mlp.mlp.fc2.weight, blk.mlp.mlp.fc2.bias
    wqkv_b, wo_b, w1_b, w2_b = (ops.block_weight(t) for t in (wqkv, wo, w1, w2))
    ln1 = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias)
    cs = (d, 2 * d, 1.4426950408889634 / D ** 0.5)
    qkv = ops.gemm_bias_act(ln1, wqkv, bqkv, w_blocked=wqkv_b, col_scale=cs)
    q, k, v = (qkv[:, :, i * d:(i + 1) * d].view(B, S, H, D) for i in range(3))
    ctx = ops.fa3_fwd(q, k, v, causal=True, k_prescaled=True, out_blocked=True)
    att = ops.gemm_bias_act(ctx, wo, bo, residual=x, w_blocked=wo_b, x_blocked_shape=(B, S, d))
    ln2 = ops.layernorm(att, blk.ln_2.weight, blk.ln_2.bias)
    o3, o1 = torch.empty_like(qkv), torch.empty_like(att)
    rows.append(measure(sm, "attention fa3_fwd5 causal", lambda: ops.fa3_fwd(q, k, v, causal=True, k_prescaled=True, out_blocked=True), a.seconds))
    rows.append(measure(sm, "qkv gemm8w", lambda: ops.gemm_bias_act(ln1, wqkv, bqkv, out=o3, w_blocked=wqkv_b, col_scale=cs), a.seconds))
    rows.append(measure(sm, "fused_mlp gelu (2 launches)", lambda: ops.fused_mlp(ln2, w1, b1, w2, b2, "gelu", residual=att, fc1_blocked=w1_b, fc2_blocked=w2_b), a.seconds))
    rows.append(measure(sm, "out-proj gemm8w+res", lambda: ops.gemm_bias_act(ctx, wo, bo, residual=x, out=o1, w_blocked=wo_b, x_blocked_shape=(B, S, d)), a.seconds))
    rows.append(measure(sm, "layernorm", lambda: ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias), a.seconds))
    z = torch.zeros_like(ln1)
    rows.append(measure(sm, "qkv gemm8w, all-zero activations", lambda: ops.gemm_bias_act(z, wqkv, bqkv, out=o3, w_blocked=wqkv_b, col_scale=cs), a.seconds))
    res["rows"] = rows
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
