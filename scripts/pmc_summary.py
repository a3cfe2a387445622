"""Per-kernel sums of rocprofv3 --pmc counters from the rocpd sqlite output (view counters_collection).
usage: python scripts/pmc_summary.py <results.db> [name-substring ...]"""
import re
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection "
                        "group by kernel_name, counter_name order by sum(value) desc"))
pats = sys.argv[2:]
for name, counter, n, total in rows[:60]:
    short = re.sub(r"\(anonymous namespace\)::", "", name)
    short = re.sub(r"^void ", "", short).split("(")[0][:70]
    if pats and not any(p in short for p in pats):
        continue
    print(f"{short:72s} {counter:12s} dispatches {n:7d}  sum {total:.6g}  per-dispatch {total / n:.6g}")
