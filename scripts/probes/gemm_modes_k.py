"""Per-round time of the 256-tile GEMM against K for the three operand modes at one M x N (5632 x 3584: 22 x 14 = 308 tiles):
slope = the K loop, intercept = prologue + epilogue.  NT: A[M,K] B[N,K]; dX form: B stored [K,N]; dW form: both contraction-major,
fp32 read-modify-write output."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, N = 10752, 3584          # 42 x 14 = 588 tiles = 2.30 rounds
for mode in ("nt", "dx", "dw"):
    ts = []
    for Kd in (256, 2048, 8192, 16384):
        if mode == "nt":
            a, b = torch.randn(M, Kd, device=dev).bfloat16(), torch.randn(N, Kd, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            run = lambda: K.gemm_nt(a, b, out=out)
        elif mode == "dx":
            a, b = torch.randn(M, Kd, device=dev).bfloat16(), torch.randn(Kd, N, device=dev).bfloat16()
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            run = lambda: K.gemm(a, b, trans_b=True, out=out)
        else:
            a, b = torch.randn(Kd, M, device=dev).bfloat16(), torch.randn(Kd, N, device=dev).bfloat16()
            out = torch.zeros(M, N, device=dev, dtype=torch.float32)
            run = lambda: K.gemm(a, b, trans_a=True, trans_b=True, out=out, residual=out)
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 5 * 1e-3
        ts.append((Kd, t))
        print(f"  {mode} K={Kd:6d}: {t*1e6:8.1f} us  {2*M*N*Kd/t/1e12:7.1f} TF/s", flush=True)
    (k0, t0), (k1, t1) = ts[1], ts[-1]
    slope = (t1 - t0) / (k1 - k0)
    print(f"  {mode}: slope {slope*1e9*64:.1f} ns per 64-wide K tile per launch ({2*M*N/slope/1e12:.0f} TF/s in the K loop), intercept {1e6*(t0 - slope*k0):.1f} us")
