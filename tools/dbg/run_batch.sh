python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest4.log 2>&1; tail -6 gpurun_out/r3_gputest4.log
python bench.py > gpurun_out/r3_bench4.json 2> gpurun_out/r3_bench4.err; python -c "
import json; d=json.load(open('gpurun_out/r3_bench4.json')); print(d['value'], d['ms_per_step'], d['mfma_roofline_frac_end_to_end']); print(json.dumps(d['kernels_per_layer'])); print(d['extra']['c5']); print(d['extra']['swiglu']); print(d['extra']['attention_functional'])"
