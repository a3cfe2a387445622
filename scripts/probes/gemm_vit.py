"""ViT-shaped GEMMs under both tiles (SPACER_GEMM_TILE forces the kernel): is the cost model's choice right?"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
shapes = [(4160, 5120, 1280), (4160, 1280, 5120), (4160, 3840, 1280), (4160, 1280, 1280), (1280, 5120, 4160), (5120, 1280, 4160), (3840, 1280, 4160),
          (1280, 1280, 4160), (1040, 5120, 5120), (1040, 3584, 5120)]
for M, N, Kd in shapes:
    a = [torch.randn(M, Kd, device=dev).bfloat16() for _ in range(4)]; b = [torch.randn(N, Kd, device=dev).bfloat16() for _ in range(4)]
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3): K.gemm_nt(a[0], b[0], out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(20): K.gemm_nt(a[i % 4], b[i % 4], out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    print(f"  {M:6d} {N:6d} {Kd:6d}: tile {K._lib.load().spacer_gemm_tile(M, N, Kd, 1, None)}  {2 * M * N * Kd / t / 1e12:7.1f} TF/s {t * 1e6:8.1f} us")
