#!/bin/bash
# Runs ON the GPU box (via gpurun): rocprofv3 kernel-trace stats of the benchmark command and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE separately: they do not fit one pass on gfx950) for the HBM traffic of the dominant kernel.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_${1:-r01}
rm -rf $OUT
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python bench.py --steps 3 --warmup 1 > $OUT/bench_stats.json 2> $OUT/bench_stats.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o k -- python bench.py --steps 1 --warmup 1 --no-extra > /dev/null 2> $OUT/pmc_fetch.err
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o k -- python bench.py --steps 1 --warmup 1 --no-extra > /dev/null 2> $OUT/pmc_write.err
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/pmc_sq -o k -- python bench.py --steps 1 --warmup 1 --no-extra > /dev/null 2> $OUT/pmc_sq.err
ls -R $OUT | head -30
