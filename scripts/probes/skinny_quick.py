"""Packed skinny GEMM timing for the decode shapes (weights rotated through > 256 MB so the Infinity Cache is cold)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
shapes = [(64, 4608, 3584), (64, 3584, 3584), (64, 3584, 18944), (64, 152064, 3584), (64, 37888, 3584)] if len(sys.argv) < 2 else [tuple(int(x) for x in sys.argv[1:4])]
for M, N, Kd in shapes:
    nw = max(2, int(1.2e9 // (N * Kd * 2)))
    sw = N == 37888
    ws = [(K.pack_weight_frag_swiglu if sw else K.pack_weight_frag)((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(nw)]
    a = torch.randn(M, Kd, device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev)
    y = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    f = (lambda i: K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)) if sw else (lambda i: K.gemm_skinny_packed_acc(a, ws[i], c, N))
    for i in range(nw): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 4
    e0.record()
    for r in range(reps):
        for i in range(nw): f(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (reps * nw) * 1e-3
    print(f"  {M} {N} {Kd}{' swiglu' if sw else ''}: {t*1e6:7.1f} us  {N*Kd*2/t/1e12:5.2f} TB/s")
print("-- 128 rows")
for M, N, Kd in [(128, 4608, 3584), (128, 3584, 18944), (128, 152064, 3584), (128, 37888, 3584)]:
    nw = max(2, int(1.2e9 // (N * Kd * 2)))
    sw = N == 37888
    ws = [(K.pack_weight_frag_swiglu if sw else K.pack_weight_frag)((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(nw)]
    a = torch.randn(M, Kd, device=dev).bfloat16()
    c = torch.zeros(M, N, device=dev)
    y = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    f = (lambda i: K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)) if sw else (lambda i: K.gemm_skinny_packed_acc(a, ws[i], c, N))
    for i in range(nw): f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(4):
        for i in range(nw): f(i)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (4 * nw) * 1e-3
    print(f"  {M} {N} {Kd}{' swiglu' if sw else ''}: {t*1e6:7.1f} us  {N*Kd*2/t/1e12:5.2f} TB/s")
