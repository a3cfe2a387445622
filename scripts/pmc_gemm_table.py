"""HBM-side traffic of the 256-tile GEMM per shape from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE -- they do not fit one
pass) over scripts/gemm_step_shapes.py.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md HBM section: on gfx950
FETCH_SIZE reports half of the bytes of wide streaming reads; both counters are in KiB); the split-K reduce launch that follows a
GEMM launch is charged to it; averages are launch-weighted over the cfg3 step's mix, per kernel instantiation.

    collect(kinds)            run both passes (subprocess rocprofv3) and return the table -- bench.py's live `roofline.traffic`
    pmc_gemm_table.py out.json [kind ...]     the same from the command line, written to out.json
"""
import json
import os
import sqlite3
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from gemm_step_shapes import SHAPES  # noqa: E402

def inst(kind: str, f32: int) -> str:
    """rocprof name of the instantiation <BALANCED, TA, TB, STG16> a step shape runs on (STG16 = bf16 output, no residual)."""
    if kind == "swiglu" or (kind == "nt" and not f32):
        return "gemm_bf16_nt_256h_kernel<true, false, false, true>"
    if kind == "nt":
        return "gemm_bf16_nt_256h_kernel<true, false, false, false>"
    if kind == "dx":
        return "gemm_bf16_nt_256h_kernel<true, false, true, false>" if f32 else "gemm_bf16_nt_256h_kernel<true, false, true, true>"
    return "gemm_bf16_nt_256h_kernel<true, true, true, false>"


def _per_dispatch(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? and "
                            "(kernel_name like '%gemm_bf16_nt_%' or kernel_name like '%gemm_tail_re%' or kernel_name like '%gemm_tail_ro%') "
                            "order by dispatch_id", (counter,)))
    main, extra, pending = [], {}, 0.0
    for _, name, v in rows:
        if "tail_rows" in name:                 # K-tail copies run BEFORE the dW launch they serve
            pending += v
        elif "tail_reduce" in name:
            extra[len(main) - 1] = extra.get(len(main) - 1, 0.0) + v
        else:
            main.append(v + pending)
            pending = 0.0
    return main, extra


def _pass(counter, reps, kinds, tmp, timeout):
    out = os.path.join(tmp, counter)
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", out, "-o", "r", "--", sys.executable,
           os.path.join(HERE, "gemm_step_shapes.py"), str(reps)] + sorted(kinds or [])
    subprocess.run(cmd, check=True, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   env=dict(os.environ, TMPDIR=tmp), cwd=tmp)
    for root, _, files in os.walk(out):
        for f in files:
            if f.endswith(".db"):
                return os.path.join(root, f)
    raise FileNotFoundError("rocprofv3 left no .db")


def collect(kinds=None, reps=2, timeout=420):
    shapes = [s for s in SHAPES if not kinds or s[0] in kinds]
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        (fetch, fx), (write, wx) = (_per_dispatch(_pass(c, reps, kinds, tmp, timeout), c) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    assert len(fetch) == len(write) == reps * len(shapes), (len(fetch), len(write), len(shapes))
    table, tot = [], {}
    for i, (kind, M, N, K, launches, f32, acc) in enumerate(shapes):
        idx = range(i * reps, (i + 1) * reps)
        f = sum(fetch[k] + fx.get(k, 0.0) for k in idx) / reps
        w = sum(write[k] + wx.get(k, 0.0) for k in idx) / reps
        hbm = (2.0 * f + w) * 1024.0
        out_b = (2.0 * M * (N // 2) if kind == "swiglu" else (4 if f32 else 2) * M * N)
        algo = 2.0 * (M * K + N * K) + out_b + ((4.0 * M * N) if acc else 0.0)
        table.append(dict(kind=kind, M=M, N=N, K=K, launches_per_step=launches, fetch_kib=f, write_kib=w, hbm_bytes=hbm,
                          algorithmic_bytes=algo, ratio=hbm / algo))
        t = tot.setdefault(inst(kind, f32), [0.0, 0, 0.0])
        t[0] += hbm * launches; t[1] += launches; t[2] += algo * launches
    return dict(method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over scripts/gemm_step_shapes.py; "
                       "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024; launch-weighted over the cfg3 step's shapes",
                per_kernel={k: dict(hbm_bytes_per_launch=v[0] / max(1, v[1]), algorithmic_bytes_per_launch=v[2] / max(1, v[1]),
                                    launches_per_step=v[1]) for k, v in tot.items()}, shapes=table)


if __name__ == "__main__":
    res = collect(set(sys.argv[2:]) or None)
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res["per_kernel"]))
