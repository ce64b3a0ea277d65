#!/usr/bin/env python3
"""In-model duration of each launch of a layer, from a rocprofv3 --kernel-trace CSV of bench.py (the forward with the LayerNorms
folded: per layer QKV, attention, out-proj, fc1, fc2 = GEMM, attention, GEMM, GEMM, GEMM).  rocprofv3's begin/end stamps of
back-to-back launches touch, so a duration includes the launch boundary behind the previous kernel.
    python tools/trace_inmodel.py gpurun_out/prof_r03/trace/bench_kernel_trace.csv"""
import csv, sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
kind = lambda n: "L" if "layernorm" in n else "A" if "fa3_fwd5" in n else "G" if "gemm8w" in n else "x"
s = "".join(kind(r[2]) for r in rows)
pat, names = "GAGGG", ["qkv", "attention", "out-proj", "fc1", "fc2"]
reps = 22
acc, steps, i = defaultdict(list), [], 0
while True:
    j = s.find(pat * reps, i)
    if j < 0:
        break
    for l in range(reps):
        for p_ in range(5):
            r = rows[j + l * 5 + p_]
            acc[p_].append(r[1] - r[0])
    steps.append((rows[j + reps * 5 - 1][1] - rows[j][0]) / reps)
    i = j + len(pat) * reps
print("runs of", reps, "folded layers found:", len(steps))
tot = 0.0
for p_ in range(5):
    v = acc[p_]
    a = sum(v) / len(v)
    tot += a
    print(f"{names[p_]:10s} avg {a / 1e3:8.1f} us   min {min(v) / 1e3:8.1f}   max {max(v) / 1e3:8.1f}")
print(f"sum per layer {tot / 1e3:.1f} us; wall per layer {sum(steps) / len(steps) / 1e3:.1f} us")
