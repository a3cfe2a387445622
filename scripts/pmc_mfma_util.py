"""MFMA pipe utilisation per kernel from one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES pass (rocpd sqlite).
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all 1024 SIMDs, SQ_BUSY_CYCLES those of the 32 shader engines' SQs
(8 XCD x 4), so util = (MFMA / 1024) / (BUSY / 32).   usage: pmc_mfma_util.py <results.db> <out.md> <title> [substr ...]"""
import re
import sqlite3
import sys

db, out, title = sys.argv[1:4]
pats = sys.argv[4:]
cur = sqlite3.connect(db).cursor()
agg = {}
for name, counter, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0][:80]
    if pats and not any(p in short for p in pats):
        continue
    a = agg.setdefault(short, {"n": 0})
    a[counter] = a.get(counter, 0.0) + val
    if counter == "SQ_BUSY_CYCLES":
        a["n"] += 1
with open(out, "w") as f:
    f.write(f"# {title}\n\nrocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; MFMA utilisation = (MFMA busy / 1024 SIMDs) / (SQ busy / 32 SQs)\n\n")
    f.write("| kernel | launches | MFMA-busy cycles per SIMD / launch | busy cycles / launch | MFMA utilisation |\n|---|---:|---:|---:|---:|\n")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0)):
        m, b, n = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024, a.get("SQ_BUSY_CYCLES", 0.0) / 32, max(1, a["n"])
        if b > 0:
            f.write(f"| `{k}` | {n} | {m / n:,.0f} | {b / n:,.0f} | {m / b:.3f} |\n")
print(open(out).read())
