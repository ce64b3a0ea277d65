#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, grouped by (previous kernel, next kernel).
    python tools/trace_gaps.py gpurun_out/prof_r03/trace/bench_kernel_trace.csv
Only pairs whose gap is < 200 us are counted (larger gaps are host-side: synchronisations between bench legs)."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"(_Z\d+)?([A-Za-z_0-9]+)", n)
    base = m.group(2) if m else n[:30]
    if "gemm8w" in n:
        base += "/res" if ("Lb1E" in n or ", true, 0>" in n or "E, true" in n) else ""
        base += "/" + re.sub(r"[^0-9A-Za-z]", "", n[-40:])[-14:]
    return base[:60]


rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps = defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 200_000:
        gaps[(short(n0), short(n1))].append(g)
tot = 0
print(f"{'previous -> next':100s} {'n':>6s} {'avg ns':>9s} {'min':>7s} {'p50':>7s} {'max':>8s}")
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 20:
        continue
    v.sort()
    tot += sum(v)
    print(f"{(k[0] + ' -> ' + k[1]):100s} {len(v):6d} {sum(v) / len(v):9.0f} {v[0]:7d} {v[len(v) // 2]:7d} {v[-1]:8d}")
print("total idle ns in listed pairs:", tot)
