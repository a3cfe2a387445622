"""256-tile GEMM launch time at the cfg3 two-group shapes (bf16 out: gate|up width and q|k|v width; fp32 + residual: o / down widths),
K = 128 (all fixed cost) and the real K: microseconds per round of 256 tiles."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for M, N, Kd, f32 in [(10996, 37888, 128, False), (10996, 37888, 3584, False), (10996, 4608, 3584, False), (10996, 3584, 128, True), (10996, 3584, 3584, True),
                      (10996, 3584, 18944, True)]:
    a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
    out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    run = (lambda: K.gemm_nt(a, b, out=out, residual=out)) if f32 else (lambda: K.gemm_nt(a, b, out=out))
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5 * 1e-3
    rounds = -(-M // 256) * -(-N // 256) / 256
    print(f"  {M:6d} {N:6d} K={Kd:5d} {'f32+resid' if f32 else 'bf16     '}: {t*1e6:8.1f} us  {2*M*N*Kd/t/1e12:7.1f} TF/s  us/round {t*1e6/rounds:.1f}")
