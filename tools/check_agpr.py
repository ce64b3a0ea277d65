#!/usr/bin/env python3
"""Build-time check for kernels that own accumulator registers through inline asm: no COMPILER-generated instruction
(anything outside #ASMSTART/#ASMEND) may touch an accumulator register >= the given floor.
usage: tools/check_agpr.py <isa.s> <floor>"""
import re, sys
path, floor = sys.argv[1], int(sys.argv[2])
inasm, bad, used = False, [], set()
for n, l in enumerate(open(path), 1):
    if "ASMSTART" in l: inasm = True; continue
    if "ASMEND" in l: inasm = False; continue
    x = l.split(";")[0]
    if inasm or not x.strip() or x.strip().startswith("."): continue
    regs = [int(m) for m in re.findall(r"\ba(\d+)\b", x)]
    for m in re.finditer(r"\ba\[(\d+):(\d+)\]", x):
        regs += list(range(int(m.group(1)), int(m.group(2)) + 1))
    used.update(regs)
    if any(r >= floor for r in regs): bad.append((n, x.strip()))
print(f"compiler-generated code uses accumulator registers {sorted(used)[:1]}..{sorted(used)[-1:]} ({len(used)} distinct); floor {floor}")
for n, x in bad[:10]: print("  VIOLATION line", n, x)
sys.exit(1 if bad else 0)
