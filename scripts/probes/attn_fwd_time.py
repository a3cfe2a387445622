"""cfg3-layout scoring attention forward (1 and 2 groups per pass) and the ViT per-frame attention: time and TF/s, the three forward kernels
(SPACER_ATTN_FWD = reg | pipe), plus a bit-compare against the register-staged one."""
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(3): fn()
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
    flops = 0
    for qs, ql, ps, pl in seg_list:
        flops += 4.0 * D * Hq * (ql * pl + (ql * (ql + 1) / 2 if causal else ql * ql))
    res = {}
    for mode in ("reg", "pipe"):
        os.environ["SPACER_ATTN_FWD"] = mode
        o, lse = K.attn_fwd(q, k, v, segs, mq, Hq, Hkv, D, causal, D ** -0.5)
        t = timeit(lambda: K.attn_fwd(q, k, v, segs, mq, Hq, Hkv, D, causal, D ** -0.5))
        res[mode] = (t, o.clone(), lse.clone())
    os.environ.pop("SPACER_ATTN_FWD", None)
    same = torch.equal(res["reg"][1], res["pipe"][1]) and torch.equal(res["reg"][2], res["pipe"][2])
    print(f"  {name:34s} " + "   ".join(f"{m} {res[m][0] * 1e6:7.1f} us {flops / res[m][0] / 1e12:5.0f} TF/s" for m in res)
          + f"   identical={same} o_diff={float((res['reg'][1].float() - res['pipe'][1].float()).abs().max()):.1e} lse_diff={float((res['reg'][2] - res['pipe'][2]).abs().max()):.1e}", flush=True)


P, C, Kn = 1402, 512, 8
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = one + [(5498 + s[0], s[1], 5498 + s[2] if s[3] else 0, s[3]) for s in one]
case("cfg3 scoring, 1 group", one, 28, 4, 128, True)
case("cfg3 scoring, 2 groups", two, 28, 4, 128, True)
case("prefill 8 prompts", [(i * P, P, 0, 0) for i in range(8)], 28, 4, 128, True)
case("ViT 8 frames x 520 (D=80)", [(i * 520, 520, 0, 0) for i in range(8)], 16, 16, 80, False)
case("ViT cfg5 16 frames x 1024 (D=80)", [(i * 1024, 1024, 0, 0) for i in range(16)], 16, 16, 80, False)
