"""cfg3-layout scoring attention forward+backward, a few launches: the workload the rocprofv3 --pmc probes wrap."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
P, C, Kn, Hq, Hkv, D = 1402, 512, 8, 28, 4, 128
T = P + Kn * C
qkv = (torch.randn(T, (Hq + 2 * Hkv) * D, device=dev)).bfloat16()
q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
segs = K.make_segments([(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)], dev)
d_o = torch.randn(T, Hq * D, device=dev).bfloat16()
dqkv = torch.empty_like(qkv)
dk32 = torch.zeros(T, Hkv * D, device=dev); dv32 = torch.zeros(T, Hkv * D, device=dev)
for _ in range(4):
    o, lse = K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5)
    K.attn_bwd(q, k, v, o, d_o, lse, segs, P, Hq, Hkv, D, True, D ** -0.5, dq=dqkv[:, :Hq * D], dk32=dk32, dv32=dv32)
torch.cuda.synchronize()
