#!/bin/bash
# A/B helper for the GPU box: runs bench.py variants back to back in one process environment and prints value / ms / peak memory.
# usage: scripts/ab_bench.sh "<label>|<bench args>" ...
mkdir -p gpurun_out
for spec in "$@"; do
  label="${spec%%|*}"; args="${spec#*|}"
  python bench.py --no-variants --no-cpu-baseline --no-pmc $args 2>gpurun_out/ab_$label.err | grep "^{" | tail -1 > gpurun_out/ab_$label.json
  python - "$label" <<'PY'
import json, sys
lab = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/ab_{lab}.json"))
    k = d.get("kernels", {})
    dw = k.get("gemm_bf16_nt_256h_kernel<true, true, true, false>", {}).get("tflops")
    print(f"{lab:28s} {d['value']:8.3f} samples/s  {d['ms_per_step']:9.1f} ms  peak {d['hbm_peak_gb']:6.1f} GB  dom {d['roofline']['achieved']:7.1f} TF/s  dW {dw}  decode {d.get('decode',{}).get('ms_per_token_step')}")
except Exception as e:
    print(lab, "FAILED", e)
    print(open(f"gpurun_out/ab_{lab}.err").read()[-600:])
PY
done
