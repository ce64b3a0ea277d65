#!/bin/bash
# Runs ON the GPU box: SQ counters + clock of the attention structures (fa_pmc.py modes; separate passes; no trace domains besides kernel-trace).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/fa_pmc
rm -rf $OUT; mkdir -p $OUT
for impl in 5 5p 3; do
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a$impl -o k -- python3 tools/fa_pmc.py $impl 300 > $OUT/a$impl.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $OUT/b$impl -o k -- python3 tools/fa_pmc.py $impl 300 > $OUT/b$impl.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/fa_pmc/*/")):
    acc = collections.defaultdict(list)
    for p in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if "fa3_fwd" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    dur = []
    for p in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if "fa3_fwd" in row["Kernel_Name"]:
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
    dur = dur[len(dur) // 2:]
    print(d, "dispatches", len(dur), "mean us", sum(dur) / max(1, len(dur)))
    for c, v in sorted(acc.items()):
        v = v[len(v) // 2:]
        print(f"   {c:28s} {sum(v) / len(v):.4g}")
PY
