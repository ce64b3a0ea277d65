#!/usr/bin/env python3
"""ISA scan for the store-data hazard found in round 3: a vector-memory STORE whose data registers are written
again by one of the next few instructions (hipcc reuses dead registers at once; on MI355X, with several waves storing, the store
had not fetched lanes 12..15 / 44..47 of its second dword yet).  Reports every store followed within WINDOW instructions by an
instruction whose destination overlaps the store's data (or address) registers.
usage: tools/check_store_war.py <isa.s> [kernel-name-substring] [window]"""
import re, sys

path = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
WINDOW = int(sys.argv[3]) if len(sys.argv) > 3 else 8


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


STORE = re.compile(r"^\s*(buffer_store_\w+|global_store_\w+|flat_store_\w+|scratch_store_\w+)\s+(.*)")
cur, lines, bad, nstores = None, [], [], 0
for raw in open(path):
    m = re.match(r"^(_Z\w+|\w+_kernel\w*):", raw)
    if m:
        cur, lines = m.group(1), []
        continue
    if cur is None or (pat and pat not in cur):
        continue
    x = raw.split(";")[0].rstrip()
    if not x.startswith("\t") or x.strip().startswith("."):
        continue
    lines.append(x.strip())
    if "s_endpgm" in x:
        for i, l in enumerate(lines):
            ms = STORE.match(l)
            if not ms:
                continue
            nstores += 1
            ops = [o.strip() for o in ms.group(2).split(",")]
            # global_store: vaddr, vdata, saddr ; buffer_store: vdata, vaddr, srsrc, soffset
            data = regs(ops[1]) if ms.group(1).startswith(("global_", "flat_", "scratch_")) else regs(ops[0])
            for j in range(i + 1, min(i + 1 + WINDOW, len(lines))):
                n = lines[j]
                if n.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "ds_store", ";")):
                    continue
                parts = n.split(None, 1)
                if len(parts) < 2:
                    continue
                dst = regs(parts[1].split(",")[0])
                if n.startswith("v_permlane") or "swap" in parts[0]:
                    dst |= regs(parts[1].split(",")[1]) if "," in parts[1] else set()
                if dst & data:
                    bad.append((cur, l, j - i, n))
                    break
        cur = None
print(f"{nstores} stores scanned, {len(bad)} with their data registers rewritten within {WINDOW} instructions")
for k, st, d, n in bad[:40]:
    print(f"  {k[:60]}: `{st[:70]}` -> +{d}: `{n[:70]}`")
sys.exit(1 if bad else 0)
