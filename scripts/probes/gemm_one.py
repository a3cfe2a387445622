"""One GEMM shape, a few launches: the workload the rocprofv3 --pmc probes wrap (SPACER_GEMM_TILE picks the kernel)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, N, Kd = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 8192, 8192)
a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
for _ in range(6):
    K.gemm_nt(a, b, out=out)
torch.cuda.synchronize()
