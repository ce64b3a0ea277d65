python -m pytest tests -m gpu -x -q > gpurun_out/r3_gputest5.log 2>&1; tail -5 gpurun_out/r3_gputest5.log
bash tools/collect_profiles.sh r03 > gpurun_out/collect_r03.log 2>&1; tail -3 gpurun_out/collect_r03.log
