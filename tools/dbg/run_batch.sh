python tools/dbg/fa5_store_ab.py 2>&1 | tail -5
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest3.log 2>&1; tail -4 gpurun_out/r3_gputest3.log
for i in 1 2; do python bench.py --no-extra --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step', round(d['ms_per_step'],3))"; done
