"""The precise mode's gate|up projection at the cfg3 two-group shape (10 996 x 37 888 x 3584, fp32 output) as ONE launch over the
K-concatenated pair operands or as two accumulate passes: the workload of the round-4 PMC comparison (FETCH_SIZE / WRITE_SIZE per
mode) and a wall-time A/B.      python scripts/probes/pair_gemm_one.py fused|twopass [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacer_amd import kernels as K   # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fused"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
M, N, Kd = 10996, 37888, 3584
g = torch.Generator(device=dev).manual_seed(1)
a = torch.randn(M, Kd, device=dev, generator=g) * 0.5
hi, lo = K.split_pair(a)
ws = [(torch.randn(N, Kd, device=dev, generator=g) * 0.02).to(torch.bfloat16) for _ in range(2)]
out = torch.empty(M, N, device=dev, dtype=torch.float32)
K.PAIR_TWOPASS = mode == "twopass"
K.gemm_pair(hi, lo, ws[0], out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(reps):
    K.gemm_pair(hi, lo, ws[r % 2], out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{mode}: {dt * 1e3:.3f} ms per pair GEMM = {4.0 * M * N * Kd / dt / 1e12:.0f} TF/s over both passes' flops")
