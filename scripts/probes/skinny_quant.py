"""Is the gate|up decode GEMM limited by block quantisation (592 workgroups on 512 slots)?  Same K, nearby N."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, Kd = 64, 3584
for N in (32768, 37888, 49152, 65536, 98304):
    nw = max(2, int(1.5e9 // (N * Kd * 2)))
    ws = [K.pack_weight_frag_swiglu((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(nw)]
    a = torch.randn(M, Kd, device=dev).bfloat16()
    y = torch.empty(M, N // 2, device=dev, dtype=torch.bfloat16)
    for i in range(nw): K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(6):
        for i in range(nw): K.gemm_skinny_swiglu(a, ws[i], N // 2, out=y)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / (6 * nw) * 1e-3
    print(f"  N={N:6d} ({N // 64:5d} workgroups): {t*1e6:7.1f} us  {N*Kd*2/t/1e12:5.2f} TB/s")
