"""gate|up decode GEMM time vs number of rows M: does re-staging A (M x K bf16 per 64-column workgroup, from L2) cost like HBM bytes?"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
N, Kd = 37888, 3584
ws = [K.pack_weight_frag_swiglu((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(5)]
for M in (8, 16, 32, 64, 96, 128):
    a = torch.randn(M, Kd, device=dev).bfloat16()
    y = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    for i in range(5): K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(8):
        for i in range(5): K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 40 * 1e-3
    abytes = (N // 64) * M * Kd * 2
    print(f"  M={M:3d}: {t*1e6:6.1f} us   W {N*Kd*2/1e6:.0f} MB + A restaged {abytes/1e6:.0f} MB -> {(N*Kd*2+abytes)/t/1e12:5.2f} TB/s through the CUs")
