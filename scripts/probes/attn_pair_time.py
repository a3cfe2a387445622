"""The pair-operand attention forward of the precise mode on the step's layouts: DMA-staged kernel (variant 0, round 5) against the
register-staged round-3 kernel (variant 1), alternating, with a bit-compare; MFMA work = 3 x the bf16 kernel's (three of the four
hi/lo partial products of both contractions)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
    x = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev)
    hi, lo = K.split_pair(x)
    qd, kd = Hq * D, Hkv * D
    cut = lambda t: (t[:, :qd], t[:, qd:qd + kd], t[:, qd + kd:])            # noqa: E731
    (qh, kh, vh), (ql, kl, vl) = cut(hi), cut(lo)
    segs = K.make_segments(seg_list, dev)
    mq = max(s[1] for s in seg_list)
    flops = 0
    for qs, qlen, ps, pl in seg_list:
        flops += 3 * 4.0 * D * Hq * (qlen * pl + (qlen * (qlen + 1) / 2 if causal else qlen * qlen))
    res = {}
    for rep in range(2):
        for var in (1, 0):
            f = lambda: K.attn_fwd_pair((qh, ql), (kh, kl), (vh, vl), segs, mq, Hq, Hkv, D, causal, D ** -0.5, variant=var)   # noqa: E731
            o = f()
            res.setdefault(var, []).append((timeit(f), o[0].clone(), o[1].clone()))
    same = torch.equal(res[0][0][1], res[1][0][1]) and torch.equal(res[0][0][2], res[1][0][2])
    t0, t1 = min(r[0] for r in res[0]), min(r[0] for r in res[1])
    print(f"| {name} | {t1 * 1e6:.1f} us ({flops / t1 / 1e12:.0f} TF/s) | {t0 * 1e6:.1f} us ({flops / t0 / 1e12:.0f} TF/s) | {same} |", flush=True)


print("| layout | register-staged (round 3) | DMA-staged, 256-row workgroups (round 5) | bit-identical |\n|---|---:|---:|---|")
P, C, Kn = 1402, 512, 8
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = [(0, P, 0, 0), (P, P, 0, 0)] + [(2 * P + i * C, C, 0, P) for i in range(Kn)] + [(2 * P + (Kn + i) * C, C, P, P) for i in range(Kn)]
case("cfg3 scoring, 1 group (28/4 heads x 128)", one, 28, 4, 128, True)
case("cfg3 scoring, 2 groups, prompts-first layout", two, 28, 4, 128, True)
case("ViT 16 frames x 520 patches (16 heads x 80)", [(i * 520, 520, 0, 0) for i in range(16)], 16, 16, 80, False)
case("ViT cfg5 16 frames x 1024 (16 heads x 80)", [(i * 1024, 1024, 0, 0) for i in range(16)], 16, 16, 80, False)
