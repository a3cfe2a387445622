import torch, sys
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, Kd in [(4096, 4096, 4096), (8192, 8192, 8192), (5496, 3584, 18944)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): K.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): K.gemm_nt(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"  {M} {N} {Kd}: {2*M*N*Kd/t/1e12:8.1f} TF/s")
