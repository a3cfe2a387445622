"""Does Infinity-Cache residency help the small decode GEMMs?  Same kernel, weights rotated through 2 / 4 / many copies."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, Kd in [(64, 4608, 3584), (64, 3584, 3584), (64, 3584, 18944)]:
    for nw in (1, 2, 4, 40):
        ws = [K.pack_weight_frag((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(nw)]
        a = torch.randn(M, Kd, device=dev).bfloat16()
        c = torch.zeros(M, N, device=dev)
        for i in range(nw): K.gemm_skinny_packed_acc(a, ws[i], c, N)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(1, 200 // nw)
        e0.record()
        for r in range(reps):
            for i in range(nw): K.gemm_skinny_packed_acc(a, ws[i], c, N)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / (reps * nw) * 1e-3
        print(f"  {M} {N} {Kd} copies={nw:3d} ({nw*N*Kd*2/1e6:7.0f} MB): {t*1e6:7.1f} us  {N*Kd*2/t/1e12:5.2f} TB/s")
        del ws
