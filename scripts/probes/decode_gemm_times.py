"""The four weight-streaming GEMMs of a 7B decode layer at 64 rows (packed weights, rotating copies > the Infinity Cache), back to back
with one event pair per launch: q|k|v (norm-folded), o, gate|up + SwiGLU, down, and the lm_head.   python scripts/probes/decode_gemm_times.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacer_amd import kernels as K   # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
FLAGS = [int(f) for f in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0]
dev = torch.device("cuda:0")
H, I, QKV, V = 3584, 18944, 4608, 152064
NC = 10


def timed(fn, n=60):
    out = []
    for rep in range(2):                                   # flags alternate twice: drift of the box shows as a difference between repeats
        for fl in FLAGS:
            K.PLAN.skinny_skew = fl
            for i in range(NC):
                fn(i)
            torch.cuda.synchronize()
            ev = []
            for r in range(n):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(r); e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
            out.append((fl, t[len(t) // 2], sum(t) / len(t)))
    K.PLAN.skinny_skew = 0
    return out


def mk(n, k):
    return [K.pack_weight_frag((torch.randn(n, k, device=dev) * 0.02).bfloat16()) for _ in range(NC)]


x32 = torch.randn(rows, H, device=dev)
xb = x32.bfloat16()
acc = torch.zeros(rows, QKV, device=dev)
rowss = torch.zeros(rows, device=dev)
w = mk(QKV, H)
res = {}
if rows <= 64:                                             # (the norm-folded projection takes <= 64 rows)
    res["q|k|v (norm-folded, 33.0 MB)"] = timed(lambda i: K.gemm_skinny_packed_normed(x32, w[i % NC], acc, rowss, QKV))
res["q|k|v (bf16 A, 33.0 MB)"] = timed(lambda i: K.gemm_skinny_packed_acc(xb, w[i % NC], acc, QKV))
del w
w = mk(H, H)
res["o (25.7 MB)"] = timed(lambda i: K.gemm_skinny_packed_acc(xb, w[i % NC], x32, H))
del w
w = [K.pack_weight_frag_swiglu((torch.randn(2 * I, H, device=dev) * 0.02).bfloat16()) for _ in range(4)]
a = torch.empty(rows, I, device=dev, dtype=torch.bfloat16)
res["gate|up + SwiGLU (271.6 MB)"] = timed(lambda i: K.gemm_skinny_swiglu(xb, w[i % 4], I, out=a))
del w
w = [K.pack_weight_frag((torch.randn(H, I, device=dev) * 0.02).bfloat16()) for _ in range(6)]
res["down (135.8 MB)"] = timed(lambda i: K.gemm_skinny_packed_acc(a, w[i % 6], x32, H))
del w
w = [K.pack_weight_frag((torch.randn(V, H, device=dev) * 0.02).bfloat16()) for _ in range(2)]
lg = torch.empty(rows, V, device=dev)
res["lm_head (1090 MB)"] = timed(lambda i: K.gemm_skinny_packed_store(xb, w[i % 2], lg, V), n=20)
for k, rows_ in res.items():
    print(f"  {k:34s} " + "   ".join(f"skew {fl}: median {med:6.1f} mean {mean:6.1f}" for fl, med, mean in rows_), flush=True)
