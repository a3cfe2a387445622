#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench.py configuration on the GPU box; leaves gpurun_out/<label>_kernel_stats.md
# (copy into profiles/ to commit) and the bench line of the same run in gpurun_out/<label>_bench.json.
# usage: scripts/profile_step.sh <label> "<title>" <bench args ...>
set -u
label="$1"; title="$2"; shift 2
root="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
out="$root/gpurun_out/prof_$label"
rm -rf "$out"; mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out" -o r -- python "$root/bench.py" --no-cpu-baseline --no-variants --no-pmc "$@" \
    > "$root/gpurun_out/${label}_bench.json" 2> "$root/gpurun_out/${label}_bench.err"
db=$(find "$out" -name '*.db' | head -1)
if [ -z "$db" ]; then echo "profile_step: rocprofv3 left no .db for $label"; tail -5 "$root/gpurun_out/${label}_bench.err"; exit 1; fi
python "$root/scripts/rocprof_summary.py" "$db" "$root/gpurun_out/${label}_kernel_stats.md" "$title (python bench.py --no-cpu-baseline --no-variants --no-pmc $*)"
rm -rf "$out"
grep "^{" "$root/gpurun_out/${label}_bench.json" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$label', d['value'], 'samples/s', d['ms_per_step'], 'ms/step', 'decode', d.get('decode'))"
