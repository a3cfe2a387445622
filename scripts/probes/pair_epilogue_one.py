"""The precise mode's gate|up projection + SwiGLU at the cfg3 two-group shape (10 996 x 37 888 x 3584) as round 4 ran it (pair GEMM ->
fp32 [T, 2I] -> swiglu_pair, with the bf16 gate|up tape) or with the producer in the pair GEMM's epilogue (round 5): workload of the PMC
comparison (FETCH_SIZE / WRITE_SIZE per kernel, scripts/pmc_summary.py) and a wall-time A/B.
    python scripts/probes/pair_epilogue_one.py fused|unfused [reps]"""
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
tape = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
K.PAIR_EPILOGUE_UNFUSED = mode == "unfused"


def once(w):
    return K.gemm_pair_swiglu(hi, lo, w, gu_out=tape)


once(ws[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in range(reps):
    once(ws[r % 2])
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(f"{mode}: {dt * 1e3:.3f} ms per gate|up + SwiGLU in the precise mode ({4.0 * M * N * Kd / dt / 1e12:.0f} TF/s over the pair GEMM's flops)")
