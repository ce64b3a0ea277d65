#!/usr/bin/env python3
"""Turn one tools/collect_profiles.sh run (gpurun_out/prof_<tag>/) into the tracked files under profiles/:
  <tag>_bench_kernel_stats.csv     rocprofv3 --kernel-trace --stats summary of `python bench.py --steps 3 --warmup 1`
  <tag>_bench_under_rocprof.json   the JSON line bench.py printed in that same run
  <tag>_pmc_sq_summary.txt         per-kernel means of the SQ counters
  pmc_traffic.json                 HBM bytes per launch per kernel (FETCH_SIZE*2 + WRITE_SIZE, KiB -> bytes), keyed by
                                   the kernel names bench.py uses
"""
import csv, glob, json, os, re, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)


ACT = {"0": "none", "1": "gelu_tanh", "2": "gelu_erf", "3": "relu", "4": "silu", "5": "swiglu"}


def short(name):
    """Kernel symbol (Itanium-mangled, or rocprofv3's partly demangled form) -> the short name bench.py uses."""
    m = re.match(r"_Z\d+(\w+?_kernel)I(.*)E+v", name)
    if m:  # mangled: template args DF16b = __bf16, DF16_ = _Float16, Li<N>E = int, Lb<0|1>E = bool
        k, rest = m.group(1), m.group(2)
        dt = "bf16" if rest.startswith("DF16b") else ("fp16" if rest.startswith("DF16_") else "?")
        ints = re.findall(r"L([ib])(\d+)E", rest)
        vals = [v for _, v in ints]
        if k in ("gemm4w16_kernel", "gemm4w16p_kernel"):
            return f"{k}<{dt},{ACT.get(vals[0], vals[0])}>"
        if k == "gemm8w_kernel":  # <T, ACT, RES, VAR, FOLD>: FOLD 1 = LayerNorm applied in the read-out, 2 = row statistics written
            fold = {"1": ",ln-fold", "2": ",ln-stats"}.get(vals[3] if len(vals) > 3 else "0", "")
            return f"{k}<{dt},{ACT.get(vals[0], vals[0])}{',residual' if vals[1] == '1' else ''}{fold}>"
        if k == "gemm_bias_act_kernel":
            return f"{k}<{dt},{ACT.get(vals[-1], vals[-1])}>"
        if k in ("fa3_fwd_kernel", "fa3_fwd2_kernel"):
            return f"{k}<{dt},D{vals[0]},{'causal' if vals[1] == '1' else 'full'}>"
        if k == "fa3_fwd5_kernel":  # <T, CAUSAL>
            return f"{k}<{dt},{'causal' if vals[0] == '1' else 'full'}>"
        if k == "fa3_fwd4_kernel":  # <T, CAUSAL, ABL, KPRE>
            return f"{k}<{dt},{'causal' if vals[0] == '1' else 'full'}{',k_prescaled' if vals[-1] == '1' and len(vals) >= 3 else ''}>"
        if k == "fa3_fwd3_kernel":  # <T, D, CAUSAL, STAMP>; the benchmark's head dim 64 keeps the short name bench.py uses
            tag = "" if vals[0] == "64" else f"D{vals[0]},"
            return f"{k}<{dt},{tag}{'causal' if vals[1] == '1' else 'full'}>"
        return f"{k}<{dt}>"
    if re.match(r"(?:void )?fa3_fwd3_kernel<bool _Accum, bool, E", name):  # <__bf16, true, false> mis-demangled
        return "fa3_fwd3_kernel<bf16,causal>"
    if re.match(r"(?:void )?fa3_fwd4_kernel<", name):  # <__bf16, true, 0, true>
        return "fa3_fwd4_kernel<bf16,causal,k_prescaled>"
    if re.match(r"(?:void )?fa3_fwd5_kernel<", name):  # the benchmark launches <__bf16, true>
        return "fa3_fwd5_kernel<bf16,causal>"
    m = re.match(r"(?:void )?gemm8w_kernel<bool _Accum, int, E(Lb0E)?(?:, (true|false))?, (\d+), (\d+)>", name)
    if m:  # rocprofv3's partly demangled <__bf16, ACT, RES, VAR, FOLD>: "E, false, 0, 1" = gelu_tanh (ACT 1), "ELb0E, 0, 0" = swiglu (ACT 5)
        fold = {"1": ",ln-fold", "2": ",ln-stats"}.get(m.group(4), "")
        act = "swiglu" if m.group(1) else "gelu_tanh"
        return f"gemm8w_kernel<bf16,{act}{',residual' if m.group(2) == 'true' else ''}{fold}>"
    m = re.match(r"(?:void )?(\w+_kernel)<bool _Accum, int, E(?:, (\d+))?", name)
    if m:  # rocprofv3 mis-demangles <__bf16, 1, ...>: only the gelu_tanh (ACT = 1) GEMMs of the benchmark show up so
        return f"{m.group(1)}<bf16,gelu_tanh>"
    return name[:60]


def counter_means(path):
    acc = defaultdict(lambda: defaultdict(list))
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name") or ""
        if "at::native" in name or "rocclr" in name or not name:
            continue
        acc[short(name)][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


stats = glob.glob(f"{src}/trace/**/*kernel_stats.csv", recursive=True)
if stats:
    shutil.copy(stats[0], f"profiles/{tag}_bench_kernel_stats.csv")
if os.path.exists(f"{src}/bench_stats.json"):
    shutil.copy(f"{src}/bench_stats.json", f"profiles/{tag}_bench_under_rocprof.json")
traffic = defaultdict(float)
for sub, ctr, mul in (("pmc_fetch", "FETCH_SIZE", 2.0), ("pmc_write", "WRITE_SIZE", 1.0)):
    for p in glob.glob(f"{src}/{sub}/**/*counter_collection.csv", recursive=True):
        for k, cs in counter_means(p).items():
            if ctr in cs:
                traffic[k] += cs[ctr] * 1024.0 * mul
if traffic:
    out = dict(sorted(traffic.items()))
    out["_note"] = ("HBM-side bytes per launch of each kernel in `bench.py --steps 1 --warmup 1 --no-extra` on one MI355X: "
                    "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (tools/collect_profiles.sh), "
                    "value*1024 bytes, FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, "
                    "MI355X_MICROARCH.md section HBM; the LayerNorm and attention rows land on their algorithmic 128 / "
                    "256 MiB with that correction). Infinity-Cache hits are counted, so the GEMM rows include tile "
                    "re-reads served on-die.  Names as in bench.py kernel_rooflines().")
    json.dump(out, open("profiles/pmc_traffic.json", "w"), indent=1)
for p in glob.glob(f"{src}/pmc_sq/**/*counter_collection.csv", recursive=True):
    with open(f"profiles/{tag}_pmc_sq_summary.txt", "w") as f:
        for k, cs in counter_means(p).items():
            f.write(k + "\n")
            for c, v in sorted(cs.items()):
                f.write(f"   {c:32s} mean={v:.4g}\n")
            if cs.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in cs:
                # matrix-pipe busy fraction: MFMA busy cycles (summed over SIMDs) / (kernel cycles x 1024 SIMDs);
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs
                busy = cs["SQ_VALU_MFMA_BUSY_CYCLES"] / (cs["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
                f.write(f"   {'=> matrix pipe busy':32s} {100 * busy:.1f} %\n")
print(open("profiles/pmc_traffic.json").read() if traffic else "no traffic")
