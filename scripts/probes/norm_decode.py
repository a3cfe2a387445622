"""Decode-step RMSNorm (64 rows x 3584, fp32 in) timed inside a hipGraph of 56 launches like one token step."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
x = torch.randn(64, 3584, device=dev); w = (torch.randn(3584, device=dev) * 0.1 + 1).bfloat16()
h = torch.empty(64, 3584, device=dev, dtype=torch.bfloat16)
def run():
    for _ in range(56): K.rmsnorm_fwd(x, w, 1e-6, out=h)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): g.replay()
e1.record(); torch.cuda.synchronize()
print(f"rmsnorm_fwd 64x3584: {e0.elapsed_time(e1) / (20 * 56) * 1e3:.2f} us per launch in a graph")
ref = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w.float()
print("max err", float((h.float() - ref).abs().max()))
