#!/bin/bash
# tools/isa.sh <hip source in csrc> <mangled-kernel-regex> [extra flags]  -> /tmp/isa_kernel.s (device ISA of one kernel)
set -e
cd /root/repo/ml-inference-optimizer_amd/csrc
src=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Wno-unused-value "$@" -S --cuda-device-only $src -o /tmp/isa_all.s
awk "/^${pat}:/,/s_endpgm/" /tmp/isa_all.s > /tmp/isa_kernel.s
wc -l /tmp/isa_kernel.s
grep -A25 "^\s*.amdhsa_kernel ${pat}" /tmp/isa_all.s | grep "next_free_vgpr\|private_segment_fixed_size\|accum_offset" || true
