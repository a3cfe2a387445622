mkdir -p gpurun_out
(timeout 1500 python -m pytest tests/test_precise_gpu.py -q > gpurun_out/r05_gputest_7.log 2>&1; tail -4 gpurun_out/r05_gputest_7.log)
python scripts/probes/attn_pair_time.py > gpurun_out/r05_attn_pair.md 2> gpurun_out/r05_attn_pair.err; cat gpurun_out/r05_attn_pair.md; tail -2 gpurun_out/r05_attn_pair.err
for v in reg dma reg dma; do
  SPACER_ATTN_PAIR=$v python bench.py --no-cpu-baseline --no-variants --no-pmc --precise-logps --steps 3 --warmup 1 2>gpurun_out/r05_precise_attn_$v.err | tail -1 > gpurun_out/r05_precise_attn_$v.json
  python -c "
import json; d=json.load(open('gpurun_out/r05_precise_attn_$v.json')); print('precise step, pair attention=$v', d['value'], d['ms_per_step'])" || tail -3 gpurun_out/r05_precise_attn_$v.err
done
python bench.py --no-cpu-baseline --no-variants --no-pmc --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fast step', d['value'], d['ms_per_step'])"
