mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_precise_gpu.py tests/test_rollout_gpu.py tests/test_configs_gpu.py tests/test_engine_gpu.py -x -q > gpurun_out/r05_gputest_2.log 2>&1; tail -5 gpurun_out/r05_gputest_2.log)
for v in off nofold on off nofold on; do
  SPACER_DECODE_SMALL=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --workload cfg4 --steps 3 --warmup 1 2>gpurun_out/r05_cfg4_$v.err | tail -1 > gpurun_out/r05_cfg4_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_cfg4_$v.json')); print('cfg4 small=$v', d['value'], d['ms_per_step'], d['decode'])" || tail -3 gpurun_out/r05_cfg4_$v.err
done
for v in off on; do
  SPACER_DECODE_SMALL=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --workload cfg2 --steps 3 --warmup 1 2>gpurun_out/r05_cfg2_$v.err | tail -1 > gpurun_out/r05_cfg2_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_cfg2_$v.json')); print('cfg2 small=$v', d['value'], d['ms_per_step'], d['decode'])" || tail -3 gpurun_out/r05_cfg2_$v.err
done
scripts/profile_step.sh r05_cfg4_v1 "cfg4 (1 group per GPU, 8 decode rows), rows16 decode kernels + norm-folded gate|up" --workload cfg4 --steps 2 --warmup 1
