mkdir -p gpurun_out
(timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_rollout_gpu.py -x -q > gpurun_out/r05_gputest_3.log 2>&1; tail -4 gpurun_out/r05_gputest_3.log)
for v in off fold fold,qkv fold,o fold,down on off fold on; do
  n=$(echo $v | tr , _)
  SPACER_DECODE_SMALL=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --workload cfg4 --steps 3 --warmup 1 2>gpurun_out/r05_cfg4_$n.err | tail -1 > gpurun_out/r05_cfg4_$n.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_cfg4_$n.json')); print('cfg4 small=$v', d['value'], d['ms_per_step'], d['decode']['ms_per_token_step'])" || tail -3 gpurun_out/r05_cfg4_$n.err
done
for v in off fold on; do
  SPACER_DECODE_SMALL=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --workload cfg2 --steps 3 --warmup 1 2>gpurun_out/r05_cfg2_$v.err | tail -1 > gpurun_out/r05_cfg2_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_cfg2_$v.json')); print('cfg2 small=$v', d['value'], d['ms_per_step'], d['decode']['ms_per_token_step'])" || tail -3 gpurun_out/r05_cfg2_$v.err
done
SPACER_DECODE_SMALL=on scripts/profile_step.sh r05_cfg4_v2 "cfg4 (8 decode rows), all small-row forms on (rows16 q|k|v / o / down + norm-folded gate|up)" --workload cfg4 --steps 2 --warmup 1
