"""What would an exactly-one-resident-round decomposition of the decode gate|up GEMM be worth?  The real problem (7B: 2I = 37 888 = 592
column groups x 14 K slices = 8288 slice units, 271.6 MB) against a problem of the same bytes that IS one round of equal workgroups on
the production kernel: 512 column groups x 16 slices (2I = 32 768, K = 4096: 268.4 MB), and the 512 x 14 baseline."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spacer_amd import kernels as K   # noqa: E402

dev = torch.device("cuda:0")
CASES = [(18944, 3584, "real: 592 groups x 14 slices (tail balance)")] + [(16384, k, f"512 groups x {k // 256} slices") for k in
                                                                             (3072, 3328, 3584, 3840, 4096, 4352, 4608, 5120)]
for I, Kd, tag in CASES:
    a = torch.randn(64, Kd, device=dev).bfloat16()
    ws = [K.pack_weight_frag_swiglu((torch.randn(2 * I, Kd, device=dev) * 0.02).bfloat16()) for _ in range(4)]     # > 256 MiB in rotation
    out = torch.empty(64, I, device=dev, dtype=torch.bfloat16)
    for i in range(4):
        K.gemm_skinny_swiglu(a, ws[i], I, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(40):
        K.gemm_skinny_swiglu(a, ws[r % 4], I, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 40 * 1e3
    mb = 2 * I * Kd * 2 / 1e6
    print(f"  {tag:48s} {mb:6.1f} MB  {us:6.1f} us  {mb / us / 1e6 * 1e6 / 1e6:6.3f} TB/s  (fragment-column stride {Kd * 32 // 1024} KB)", flush=True)
    del ws
