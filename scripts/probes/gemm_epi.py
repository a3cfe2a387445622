"""How much of a GEMM launch is prologue + epilogue?  Same M, N at shrinking K (K -> 0 extrapolates the fixed part)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, f32 in [(5498, 37888, False), (5498, 18944, False), (5498, 3584, True), (37888, 3584, True)]:
    for Kd in (128, 512, 1792, 3584, 7168):
        a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        for _ in range(2): K.gemm_nt(a, b, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): K.gemm_nt(a, b, out=out)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        print(f"  {M:6d} {N:6d} K={Kd:5d} {'f32' if f32 else 'bf16'}: {t*1e6:8.1f} us  {2*M*N*Kd/t/1e12:7.1f} TF/s  tile {K._lib.load().spacer_gemm_tile(M, N, Kd, 1, None)}")
