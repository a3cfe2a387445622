mkdir -p gpurun_out
(timeout 2700 python -m pytest tests -m gpu -q > gpurun_out/r05_gputest_5.log 2>&1; tail -6 gpurun_out/r05_gputest_5.log)
