"""gate|up + SwiGLU decode GEMM (64 rows, K = 3584): time vs width around one resident round of workgroups (512 column groups =
I 16384), with the tail balance on / off (SPACER_SKINNY_NOBALANCE)."""
import os
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, Kd = 64, 3584
a = torch.randn(M, Kd, device=dev).bfloat16()
for I in (12288, 16384, 16416, 16640, 16896, 17408, 17920, 18432, 18944, 20480, 24576, 32768):
    N = 2 * I
    ws = [K.pack_weight_frag_swiglu((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(3)]
    out = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
    res = []
    for nb in (None, "1"):
        K.PLAN.skinny_no_balance = 1 if nb else 0
        for i in range(3): K.gemm_skinny_swiglu(a, ws[i], I, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(30): K.gemm_skinny_swiglu(a, ws[r % 3], I, out=out)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    K.PLAN.skinny_no_balance = 0
    print(f"  I={I:6d} col groups {N // 64:4d}: balanced {res[0]:6.1f} us ({N * Kd * 2 / res[0] / 1e6:5.2f} TB/s)   plain {res[1]:6.1f} us ({N * Kd * 2 / res[1] / 1e6:5.2f} TB/s)", flush=True)
    del ws
