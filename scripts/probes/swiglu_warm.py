"""Does Infinity-Cache (256 MiB) residency speed up the BANDWIDTH-bound decode GEMM (gate|up + SwiGLU, 64 rows)?  The same kernel
with the weights rotated through 1 / 2 / 8 copies, at widths from MALL-resident (N = 8192: 59 MB) to the real 37 888 (272 MB);
and the real width with only a PREFIX of the weights re-read just before (the "prefetch under the preceding small kernel" idea)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
M, Kd = 64, 3584


def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


a = torch.randn(M, Kd, device=dev).bfloat16()
for I in (4096, 8192, 16384, 18944):
    N = 2 * I
    for nw in (1, 2, 8):
        ws = [K.pack_weight_frag_swiglu((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(nw)]
        out = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
        state = {"i": 0}

        def run():
            K.gemm_skinny_swiglu(a, ws[state["i"] % nw], I, out=out); state["i"] += 1
        t = timeit(run, 40)
        print(f"  I={I:6d} copies={nw} ({nw * N * Kd * 2 / 1e6:6.0f} MB resident set): {t * 1e6:7.1f} us  {N * Kd * 2 / t / 1e12:5.2f} TB/s", flush=True)
        del ws
# prefix warm-up: touch the first `mb` MB of the weights of the NEXT launch with a plain read (sum) right before it
I, N = 18944, 37888
ws = [K.pack_weight_frag_swiglu((torch.randn(N, Kd, device=dev) * 0.02).bfloat16()) for _ in range(4)]
out = torch.empty(M, I, device=dev, dtype=torch.bfloat16)
for mb in (0, 16, 32, 64, 128):
    n_el = mb * 1000000 // 2
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(24)]
    for r in range(24):
        w = ws[r % 4]
        ev[r][0].record()
        if n_el:
            w[:n_el].view(torch.int32).sum()          # a streaming read of the prefix (stand-in for prefetch blocks)
        ev[r][1].record()
        K.gemm_skinny_swiglu(a, w, I, out=out)
        ev[r][2].record()
    torch.cuda.synchronize()
    pre = sum(e[0].elapsed_time(e[1]) for e in ev[4:]) / 20 * 1e3
    gem = sum(e[1].elapsed_time(e[2]) for e in ev[4:]) / 20 * 1e3
    print(f"  prefix {mb:4d} MB touched first: touch {pre:6.1f} us, gate|up GEMM {gem:6.1f} us", flush=True)
