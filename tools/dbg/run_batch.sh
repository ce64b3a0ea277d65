for i in 1 2; do
for impl in default 4wp; do
  if [ $impl = default ]; then unset MIO_GEMM_IMPL; else export MIO_GEMM_IMPL=$impl; fi
  MIO_LIB_DBG=1 python bench.py --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys,os; d=json.loads(sys.stdin.read()); print('impl', os.environ.get('MIO_GEMM_IMPL','default'), 'ms_per_step', round(d['ms_per_step'],3), 'tok/s', int(d['value']))"
done; done
