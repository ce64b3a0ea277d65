python -m pytest tests/test_gpu_parallel.py -x -q -k "k_prescaled_module or tensor2_x or sharded_stack" > gpurun_out/r3_par.log 2>&1; tail -6 gpurun_out/r3_par.log
python tools/ln_fold_bound.py 2>&1 | tail -5
