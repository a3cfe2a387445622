mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05_gputest_4.log 2>&1; tail -4 gpurun_out/r05_gputest_4.log)
for v in fold fold,attn1 fold fold,attn1; do
  n=$(echo $v | tr , _)
  SPACER_DECODE_SMALL=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --workload cfg4 --steps 3 --warmup 1 --reuse-prefill off 2>gpurun_out/r05_cfg4_$n.err | tail -1 > gpurun_out/r05_cfg4_$n.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_cfg4_$n.json')); print('cfg4 small=$v', d['value'], d['ms_per_step'], d['decode']['ms_per_token_step'])" || tail -3 gpurun_out/r05_cfg4_$n.err
done
for w in cfg2 cfg4; do for v in off on off on; do
  python bench.py --no-cpu-baseline --no-variants --no-pmc --workload $w --steps 4 --warmup 1 --reuse-prefill $v 2>gpurun_out/r05_${w}_reuse_$v.err | tail -1 > gpurun_out/r05_${w}_reuse_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_${w}_reuse_$v.json')); print('$w reuse=$v', d['value'], d['ms_per_step'], d['hbm_peak_gb'], d['config']['prefill_tape_kept'], d['config']['prefill_tape_gb'])" || tail -3 gpurun_out/r05_${w}_reuse_$v.err
done; done
