"""rmsnorm / layernorm backward timing at the cfg3 scoring shapes."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
for rows, cols, f32 in [(5498, 3584, True), (4160, 1280, True)]:
    x = torch.randn(rows, cols, device=dev)
    w = torch.randn(cols, device=dev).bfloat16()
    dy = torch.randn(rows, cols, device=dev).bfloat16()
    rstd = torch.rand(rows, device=dev) + 0.5
    dx = torch.zeros(rows, cols, device=dev); dw = torch.zeros(cols, device=dev)
    for _ in range(3): K.rmsnorm_bwd(x, w, dy, rstd, dx, dw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): K.rmsnorm_bwd(x, w, dy, rstd, dx, dw)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    nbytes = rows * cols * (4 + 2 + 4 + 4)
    print(f"  rmsnorm_bwd {rows}x{cols}: {t*1e6:7.1f} us  {nbytes/t/1e12:5.2f} TB/s")
    # variants: no accumulate
    for acc in (True, False):
        e0.record()
        for _ in range(20): K.rmsnorm_bwd(x, w, dy, rstd, dx, dw, accumulate=acc)
        e1.record(); torch.cuda.synchronize()
        print(f"    accumulate={acc}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
rows, cols = 4160, 1280
x = torch.randn(rows, cols, device=dev); w = torch.randn(cols, device=dev).bfloat16(); dy = torch.randn(rows, cols, device=dev).bfloat16()
mean = torch.randn(rows, device=dev) * 0.1; rstd = torch.rand(rows, device=dev) + 0.5
dx = torch.zeros(rows, cols, device=dev); dw = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
for _ in range(3): K.layernorm_bwd(x, w, dy, mean, rstd, dx, dw, db)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): K.layernorm_bwd(x, w, dy, mean, rstd, dx, dw, db)
e1.record(); torch.cuda.synchronize()
print(f"  layernorm_bwd {rows}x{cols}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
for rows, cols in [(4160, 1280), (5498, 4608), (4160, 5120)]:
    dyb = torch.randn(rows, cols, device=dev).bfloat16(); dbb = torch.zeros(cols, device=dev)
    for _ in range(3): K.bias_grad_(dyb, dbb)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20): K.bias_grad_(dyb, dbb)
    e1.record(); torch.cuda.synchronize()
    print(f"  bias_grad {rows}x{cols}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
rows, cols = 4160, 1280
x = torch.randn(rows, cols, device=dev); w = torch.randn(cols, device=dev).bfloat16(); b = torch.randn(cols, device=dev).bfloat16()
mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev); y = torch.empty(rows, cols, device=dev, dtype=torch.bfloat16)
for _ in range(3): K.layernorm_fwd(x, w, b, 1e-6, mean=mean, rstd=rstd, out=y)
torch.cuda.synchronize()
e0.record()
for _ in range(20): K.layernorm_fwd(x, w, b, 1e-6, mean=mean, rstd=rstd, out=y)
e1.record(); torch.cuda.synchronize()
print(f"  layernorm_fwd {rows}x{cols}: {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us")
