#!/bin/bash
# tools/scratch_count.sh <hip source in csrc> <kernel-name substring> [extra flags]
# -> scratch (spill) instruction count per matching kernel in the gfx950 ISA
set -e
cd /root/repo/ml-inference-optimizer_amd/csrc
src=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value "$@" -S --cuda-device-only $src -o /tmp/isa_all.s
python3 - "$pat" <<'PY'
import re, sys
pat = sys.argv[1]; cur = None; cnt = {}
for l in open('/tmp/isa_all.s'):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1)
        if pat in cur: cnt.setdefault(cur, 0)
    if cur and pat in cur and re.search(r'\bscratch_', l): cnt[cur] += 1
for k, v in cnt.items(): print(v, k)
PY
