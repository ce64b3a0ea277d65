#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection CSV: per kernel name, mean of each counter per dispatch."""
import csv
import glob
import sys
from collections import defaultdict

paths = sys.argv[1:] or glob.glob("gpurun_out/**/*counter_collection.csv", recursive=True)
acc = defaultdict(lambda: defaultdict(list))
for p in paths:
    for row in csv.DictReader(open(p)):
        name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
        if "at::native" in name or "rocclr" in name:
            continue
        acc[name[:70]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} n={len(v):4d} mean={sum(v)/len(v):.4g}")
