"""Per-kernel sums of every counter of one rocprofv3 --pmc pass (rocpd sqlite).  usage: pmc_dump.py <results.db> [kernel substr ...]"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
pats = sys.argv[2:]
agg = {}
for name, counter, val in cur.execute("select kernel_name, counter_name, value from counters_collection"):
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0][:60]
    if pats and not any(p in short for p in pats):
        continue
    a = agg.setdefault(short, {})
    a[counter] = a.get(counter, 0.0) + val
    a["_n_" + counter] = a.get("_n_" + counter, 0) + 1
for k, a in agg.items():
    print(k)
    for c in sorted(x for x in a if not x.startswith("_n_")):
        n = a["_n_" + c]
        print(f"    {c:32s} {a[c] / n:18,.0f} per launch ({n} samples)")
