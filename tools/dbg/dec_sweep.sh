#!/bin/bash
# decode tuning sweep: unroll factor x target workgroups (one process each; env read once per process)
for u in 1 2 4 8; do for w in 512 1024 2048; do
  echo "U=$u WGS=$w"; MIO_LIB_DBG=1 MIO_DEC_U=$u MIO_DEC_WGS=$w timeout -k 10 100 python tools/kbench.py --what decode 2>&1 | grep "paged decode" || exit 1
done; done
