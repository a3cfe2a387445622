"""Per-kernel HBM-side bytes and bandwidth from ONE rocprofv3 run with --kernel-trace --pmc FETCH_SIZE (rocpd sqlite):
bytes = 2 * FETCH_SIZE * 1024 (gfx950: FETCH_SIZE counts half of wide reads, MI355X_MICROARCH.md HBM section), duration from
the kernel trace of the same run.   usage: pmc_kernel_bw.py <results.db> <out.md> <title> [name-substring ...]"""
import re
import sqlite3
import sys

db, out, title = sys.argv[1:4]
pats = sys.argv[4:]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select c.kernel_name, c.dispatch_id, c.value, k.end - k.start from counters_collection c join kernels k "
                        "on c.dispatch_id = k.dispatch_id where c.counter_name = 'FETCH_SIZE'"))
agg = {}
for name, _, val, dur in rows:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0][:80]
    if pats and not any(p in short for p in pats):
        continue
    a = agg.setdefault(short, [0, 0.0, 0.0])
    a[0] += 1; a[1] += 2.0 * val * 1024.0; a[2] += dur * 1e-9
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --kernel-trace --pmc FETCH_SIZE; bytes = 2 x FETCH_SIZE x 1024 (reads reaching the fabric); "
            "duration from the kernel trace of the same run (PMC collection slows the launches somewhat)\n\n")
    f.write("| kernel | launches | MB read / launch | avg us | TB/s |\n|---|---:|---:|---:|---:|\n")
    for k, (n, b, t) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
        f.write(f"| `{k}` | {n} | {b / n / 1e6:.1f} | {t / n * 1e6:.1f} | {b / t / 1e12:.2f} |\n")
print(open(out).read())
