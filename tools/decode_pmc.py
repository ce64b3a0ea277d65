#!/usr/bin/env python3
"""The two decode workloads of bench.py (extra.decode_roofline, extra.decode_gqa_roofline), a few launches each, for a
rocprofv3 --pmc FETCH_SIZE --kernel-trace pass:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_decode -o k -- python3 tools/decode_pmc.py
    python tools/decode_pmc.py --summarise gpurun_out/pmc_decode      -> profiles/r03_pmc_decode.json"""
import glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "ml-inference-optimizer_amd")):
    sys.path.insert(0, p)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv
    from collections import defaultdict
    acc = defaultdict(list)
    for pth in glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(pth)):
            if row["Counter_Name"] == "FETCH_SIZE" and "decode" in row["Kernel_Name"]:
                acc[row["Kernel_Name"].split("(")[0][:60]].append(float(row["Counter_Value"]))
    out = {k: {"launches": len(v), "fetch_bytes_per_launch": sum(v) / len(v) * 1024.0 * 2.0} for k, v in acc.items()}
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE of tools/decode_pmc.py, KiB -> bytes, doubled (gfx950 reports half the bytes of wide "
                    "coalesced reads, as in profiles/pmc_traffic.json).  Algorithmic: decode_rows_kernel workload 1 GiB = 1073741824, "
                    "decode_gqa_kernel workload 512 MiB = 536870912 (every cached K and V element once).")
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_pmc_decode.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
    sys.exit(0)

import torch
import bench
for leg in (bench.decode_leg, bench.decode_gqa_leg):
    r = leg(torch.bfloat16)
    print(r["kernel"], round(r["achieved"]), "GB/s", flush=True)
