mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_prefill_reuse_gpu.py tests/test_precise_gpu.py tests/test_multigpu_gpu.py -q > gpurun_out/r05_gputest_6.log 2>&1; tail -4 gpurun_out/r05_gputest_6.log)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_default.json').read().strip().splitlines()[-1]); print('default', d['value'], d['ms_per_step'], d['roofline']['achieved'], d.get('decode'), {k:(v.get('samples_per_s') or v.get('ms_per_token_step') or v) for k,v in d.get('variants',{}).items()}); print(d['cpu_baseline']['value'], d['cpu_baseline'].get('value_kind'))"
python bench.py --precise-logps --steps 5 --warmup 2 --no-variants --no-cpu-baseline > gpurun_out/r05_bench_precise.json 2> gpurun_out/r05_bench_precise.err; python -c "
import json; d=json.loads(open('gpurun_out/r05_bench_precise.json').read().strip().splitlines()[-1]); print('precise', d['value'], d['ms_per_step'], d['roofline'])"
scripts/profile_step.sh r05_cfg3_step "cfg3 step, round 5 final tree" --steps 1 --warmup 1
scripts/profile_step.sh r05_cfg3_precise_step "cfg3 PRECISE step (--precise-logps), round 5 final tree: producers in the pair GEMM epilogues" --precise-logps --steps 1 --warmup 1
scripts/profile_step.sh r05_cfg4_step "cfg4 (1 group per GPU, 8 decode rows = the reference script's launch shape), round 5 final tree" --workload cfg4 --steps 2 --warmup 1
scripts/profile_step.sh r05_cfg2_step "cfg2 (Qwen2-VL-2B, 8 frames, K = 4, 4 groups: 16 decode rows), round 5 final tree" --workload cfg2 --steps 2 --warmup 1
scripts/profile_step.sh r05_cfg5_step "cfg5 (7B, 32 frames @ 448^2, 8 groups), round 5 final tree" --workload cfg5 --steps 1 --warmup 1
for w in cfg3_qwen25; do python bench.py --no-cpu-baseline --no-variants --no-pmc --workload $w --steps 2 --warmup 1 2>/dev/null | tail -1 > gpurun_out/r05_$w.json; python -c "
import json; d=json.load(open('gpurun_out/r05_$w.json')); print('$w', d['value'], d['ms_per_step'])"; done
