cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for impl in default 4wp; do
  if [ $impl = default ]; then unset MIO_GEMM_IMPL; else export MIO_GEMM_IMPL=$impl; fi
  export MIO_LIB_DBG=1
  rm -rf gpurun_out/ab_$impl
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$impl -o b -- python bench.py --steps 5 --warmup 2 --no-extra > gpurun_out/ab_$impl.json 2> gpurun_out/ab_$impl.err
  f=$(find gpurun_out/ab_$impl -name "*kernel_stats.csv" | head -1)
  echo "== $impl"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"], r["Percentage"])
PY
done
