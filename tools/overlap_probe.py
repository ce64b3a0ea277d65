#!/usr/bin/env python3
"""Does co-running the two halves of the batch on two HIP streams (GEMMs of one half beside the attention of the
other, GEMM workgroup budget 128 of 256 CUs) use the power budget the sequential forward leaves (tools/power_trace.py:
1303 W average against the 1400 W cap)?

    python tools/overlap_probe.py [--seconds 3]

Forms of the C2 forward (B 8, S 4096, d 1024, L 24), wall clock over queued steps + amdsmi power:
  seq8         model(x), the shipped path
  seq4x2       the two halves one after the other on one stream (what halving the batch costs by itself)
  two/lock     two streams, same phase, GEMM budget 256 and 128
  two/offset   two streams, the second half a layer behind (its first launch is an extra MLP), budgets 256 / 128
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd"), os.path.join(ROOT, "tools")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    a = ap.parse_args()
    import torch
    from mio import _lib
    from mio.synthetic import GPT2ShapedStack
    from power_trace import Sampler

    lib = _lib.lib
    dt = torch.bfloat16
    B, S, d, H, L = 8, 4096, 1024, 16, 24
    model = GPT2ShapedStack(d, H, L, 4 * d, causal=True, precision="bf16", seed=0).to(device="cuda", dtype=dt).eval()
    torch.manual_seed(0)
    x = torch.randn(B, S, d, device="cuda", dtype=dt)
    xa, xb = x[:B // 2].contiguous(), x[B // 2:].contiguous()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    sm = Sampler()

    def seq8():
        model(x)

    def seq4x2():
        model(xa)
        model(xb)

    def two(offset, budget):
        def fn():
            lib.mio_dbg_set(7, budget)
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s2):
                hb = xb
                if offset:  # half a layer of extra work in front: the second stream's MLPs meet the first stream's attention
                    model.h[0].mlp(hb, residual=hb, pre_norm=model.h[0].ln_2)
            ha = xa
            for blk in model.h:
                with torch.cuda.stream(s1):
                    ha = blk(ha)
                with torch.cuda.stream(s2):
                    hb = blk(hb)
            with torch.cuda.stream(s1):
                ha = model.ln_f(ha)
            with torch.cuda.stream(s2):
                hb = model.ln_f(hb)
            cur.wait_stream(s1)
            cur.wait_stream(s2)
            lib.mio_dbg_set(7, 0)
            return ha, hb
        return fn

    # correctness of the two-stream form against the one-stream forward
    with torch.no_grad():
        ref = model(x)
        ha, hb = two(True, 128)()
        torch.cuda.synchronize()
        err = (torch.cat([ha, hb]).float() - ref.float()).abs().max().item()
    print("two-stream vs one-stream max|d| =", err, flush=True)

    rows = []
    forms = [("seq8", seq8), ("seq4x2", seq4x2), ("two/lock/256", two(False, 0)), ("two/lock/128", two(False, 128)),
             ("two/offset/256", two(True, 0)), ("two/offset/128", two(True, 128)), ("two/offset/192", two(True, 192)),
             ("seq8 again", seq8)]
    with torch.no_grad():
        for name, fn in forms:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); per = time.perf_counter() - t0
            n = max(3, int(a.seconds / per))
            for _ in range(max(1, int(0.3 / per))):
                fn()
            torch.cuda.synchronize()
            e0 = sm.energy(); sm.start(); t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter(); smp = sm.finish(); e1 = sm.energy()
            w = [q["w"] for q in smp if q.get("w")]
            mhz = [q["mhz"] for q in smp if q.get("mhz")]
            r = {"form": name, "ms_per_step": (t1 - t0) / n * 1e3, "steps": n, "power_w": sum(w) / len(w) if w else None,
                 "mhz": sum(mhz) / len(mhz) if mhz else None,
                 "joule_per_step": (e1 - e0) / n if (e0 is not None and e1 is not None) else None}
            if offset_extra := ("offset" in name):
                r["note"] = "includes one extra MLP launch pair per step (the offset)"
            rows.append(r)
            print(json.dumps(r), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "overlap_probe.json"), "w") as f:
        json.dump({"max_abs_diff_two_stream": err, "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
