"""Quick TF/s of a few GEMM shapes with and without the split-K tail (SPACER_GEMM_TILE picks the kernel)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
shapes = [(5498, 3584, 18944), (5498, 3584, 37888), (5498, 3584, 3584), (5498, 3584, 4608), (1402, 3584, 18944), (4160, 1280, 5120),
          (4160, 5120, 1280), (4160, 1280, 1280), (4160, 3840, 1280), (5498, 4608, 3584), (8192, 8192, 8192)]
for M, N, Kd in shapes:
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    res = []
    for split in (True, False):
        for _ in range(3): K.gemm_nt(a, b, out=out, split_k=split)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): K.gemm_nt(a, b, out=out, split_k=split)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10 * 1e-3
        res.append((K._lib.load().spacer_gemm_tile(M, N, Kd, int(split), None), 2 * M * N * Kd / t / 1e12, t * 1e6))
    print(f"  {M:6d} {N:6d} {Kd:6d}: split tile{res[0][0]} {res[0][1]:7.1f} TF/s {res[0][2]:8.1f} us | nosplit tile{res[1][0]} {res[1][1]:7.1f} TF/s {res[1][2]:8.1f} us")
