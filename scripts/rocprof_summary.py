"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite output of `--kernel-trace --stats`) results .db into a small text/CSV
summary that can be committed under profiles/.   usage: python scripts/rocprof_summary.py <results.db> <out.md> [title]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z0-9_:<>, ]+?)\(", name)
    name = m.group(1) if m else name
    return name if len(name) < 90 else name[:87] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else db
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nrocprofv3 --kernel-trace --stats; durations in microseconds; total kernel time {tot / 1e6:.3f} s\n\n")
        f.write("| kernel | calls | total_us | avg_us | % |\n|---|---:|---:|---:|---:|\n")
        for name, calls, total, avg, pct in rows[:45]:
            f.write(f"| `{short(name)}` | {calls} | {total:.0f} | {avg:.2f} | {pct:.2f} |\n")
    print("wrote", out)


if __name__ == "__main__":
    main()
