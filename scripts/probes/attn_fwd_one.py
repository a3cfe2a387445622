"""cfg3 scoring attention forward, 2 groups per pass, a few launches (the workload of the PMC probes); SPACER_ATTN_FWD picks the kernel."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
P, C, Kn, Hq, Hkv, D = 1402, 512, 8, 28, 4, 128
one = [(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)]
two = one + [(5498 + s[0], s[1], 5498 + s[2] if s[3] else 0, s[3]) for s in one]
T = 2 * 5498
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).bfloat16()
q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
segs = K.make_segments(two, dev)
for _ in range(6):
    K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5)
torch.cuda.synchronize()
