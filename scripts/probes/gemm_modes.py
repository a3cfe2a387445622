"""One GEMM shape in the three operand modes (NT plain, trans_b = dX, trans_a + trans_b = dW), a few launches each, with HIP-event
timing printed: the workload the rocprofv3 --pmc LDS-conflict probes wrap.   usage: gemm_modes.py M N K [mode ...]"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, N, Kd = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 4096, 4096)
modes = sys.argv[4:] or ["nt", "tb", "tt"]
a = torch.randn(M, Kd, device=dev).bfloat16(); b = torch.randn(N, Kd, device=dev).bfloat16()
at, bt = a.t().contiguous(), b.t().contiguous()
out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
ref = None
for mode in modes:
    fn = {"nt": lambda: K.gemm_nt(a, b, out=out), "tb": lambda: K.gemm(a, bt, trans_b=True, out=out),
          "tt": lambda: K.gemm(at, bt, trans_a=True, trans_b=True, out=out)}[mode]
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    if ref is None:
        ref = out.clone()
    print(f"{mode}: {us:9.1f} us  {2.0 * M * N * Kd / us / 1e6:8.1f} TF/s  equal_to_first={bool(torch.equal(out, ref))}", flush=True)
