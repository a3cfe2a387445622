"""Production 256 x 256 GEMM at the shapes of gemm_2wg_probe.hip (bf16 output = the persistent direct-epilogue form), same box, same run."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for (M, N, Kd) in ((4096, 4096, 4096), (4096, 4096, 8192), (4096, 4096, 16384), (11008, 4608, 3584), (10996, 4608, 3584)):
    a = (torch.rand(M, Kd, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(N, Kd, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        K.gemm_nt(a, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        K.gemm_nt(a, b, out=out)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10 * 1e-3
    print(f"production {M} x {N} x {Kd} bf16 out: {t * 1e6:8.1f} us = {2.0 * M * N * Kd / t / 1e12:7.1f} TF/s | per 64-wide K tile and round {t * 1e6 / (Kd / 64) / max(1, -(-(M // 256 + (M % 256 > 0)) * (N // 256) // 256)):.3f} us")
