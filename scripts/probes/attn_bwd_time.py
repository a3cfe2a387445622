"""Attention backward (delta + dQ + dK/dV) on the cfg3 scoring layout (1 and 2 groups per pass) and the ViT per-frame layouts:
time of the whole backward, register-staged kernels (SPACER_ATTN_BWD=reg) vs the pipelined ones, and the largest difference of
dq / dk / dv between the two (dk / dv are summed with fp32 atomics, so agreement is to rounding, not bitwise)."""
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def case(name, seg_list, Hq, Hkv, D, causal):
    T = max(s[0] + s[1] for s in seg_list)
    qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
    q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    segs = K.make_segments(seg_list, dev)
    mq = max(s[1] for s in seg_list)
    o, lse = K.attn_fwd(q, k, v, segs, mq, Hq, Hkv, D, causal, D ** -0.5)
    d_o = torch.randn_like(o)
    pairs = sum(ql * pl + (ql * (ql + 1) / 2 if causal else ql * ql) for qs, ql, ps, pl in seg_list)
    flops = 7 * 2.0 * D * Hq * pairs                     # dQ kernel: 3 products, dK/dV kernel: 4
    res = {}
    for mode in ("reg", "pipe"):
        os.environ["SPACER_ATTN_BWD"] = mode
        dqkv = torch.zeros_like(qkv)
        dk32 = torch.zeros(T, Hkv * D, device=dev); dv32 = torch.zeros(T, Hkv * D, device=dev)
        K.attn_bwd(q, k, v, o, d_o, lse, segs, mq, Hq, Hkv, D, causal, D ** -0.5, dq=dqkv[:, :Hq * D], dk32=dk32, dv32=dv32)
        out = (dqkv[:, :Hq * D].float().clone(), dk32.clone(), dv32.clone())
        t = timeit(lambda: K.attn_bwd(q, k, v, o, d_o, lse, segs, mq, Hq, Hkv, D, causal, D ** -0.5, dq=dqkv[:, :Hq * D], dk32=dk32, dv32=dv32))
        res[mode] = (t, out)
    os.environ.pop("SPACER_ATTN_BWD", None)
    diffs = [float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(res["reg"][1], res["pipe"][1])]
    print(f"  {name:34s} " + "   ".join(f"{m} {res[m][0] * 1e6:7.1f} us {flops / res[m][0] / 1e12:5.0f} TF/s" for m in res)
          + "   rel max diff dq/dk/dv " + " ".join(f"{d:.1e}" for d in diffs), flush=True)


P, C, Kn = 1402, 512, 8
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = one + [(5498 + s[0], s[1], 5498 + s[2] if s[3] else 0, s[3]) for s in one]
case("cfg3 scoring, 1 group", one, 28, 4, 128, True)
case("cfg3 scoring, 2 groups", two, 28, 4, 128, True)
case("ViT 8 frames x 520 (D=80)", [(i * 520, 520, 0, 0) for i in range(8)], 16, 16, 80, False)
case("ViT cfg5 16 frames x 1024 (D=80)", [(i * 1024, 1024, 0, 0) for i in range(16)], 16, 16, 80, False)
