set -x
python tools/dbg/gemm8_nan.py 0 2>&1 | grep -v "^  " | cut -c1-110
sed -i "s/^for (M, N, K, act, bias) in .*/for (M, N, K, act, bias) in [(8192, 8192, 512, \"gelu\", True), (4101, 4104, 1024, \"silu\", True)]:/" tools/dbg/gemm8_nan.py
python tools/dbg/gemm8_nan.py 20 2>&1 | grep -v "^  " | cut -c1-110
python tools/gemm8_ab.py 5,0,24 3 xblk 2>&1 | grep -v "^check impl 5" | tail -24
python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest2.log 2>&1; tail -5 gpurun_out/r3_gputest2.log
python bench.py > gpurun_out/r3_bench2.json 2> gpurun_out/r3_bench2.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench2.json')); print(d['value'], d['ms_per_step'], d['mfma_roofline_frac_end_to_end']); print(json.dumps(d['kernels_per_layer'])); print(d['extra']['c5'])"
