"""cfg3 scoring attention, 2 groups per pass: forward, backward (delta + dQ + dK/dV) and the precise-mode pair forward, a few launches
each -- the workload of the round-3 attention PMC / kernel-trace profiles."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
P, C, Kn, Hq, Hkv, D = 1402, 512, 8, 28, 4, 128
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = one + [(5498 + s[0], s[1], 5498 + s[2] if s[3] else 0, s[3]) for s in one]
T = 2 * 5498
qkv32 = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev)
qkv = qkv32.bfloat16()
qd, kd = Hq * D, Hkv * D
q, k, v = qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:]
segs = K.make_segments(two, dev)
hi, lo = K.split_pair(qkv32)
cut = lambda t: (t[:, :qd], t[:, qd:qd + kd], t[:, qd + kd:])
(qh, kh, vh), (ql, kl, vl) = cut(hi), cut(lo)
d_o = torch.randn(T, qd, device=dev).bfloat16()
for _ in range(4):
    o, lse = K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5)
    dq = torch.empty_like(qkv)[:, :qd]
    dk, dv = torch.zeros(T, kd, device=dev), torch.zeros(T, kd, device=dev)
    K.attn_bwd(q, k, v, o, d_o, lse, segs, P, Hq, Hkv, D, True, D ** -0.5, dq=dq, dk32=dk, dv32=dv)
    K.attn_fwd_pair((qh, ql), (kh, kl), (vh, vl), segs, P, Hq, Hkv, D, True, D ** -0.5)
torch.cuda.synchronize()
