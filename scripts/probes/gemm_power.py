"""Is the 256-tile GEMM clock/power-bound?  The same launch (8192^3 and the step's 5632x37888x3584 shape) on random operands, on
operands with few set bits (small integers), and on zeros: identical instruction stream, different switching activity.
(MI355X_MICROARCH.md "DVFS give-back": zero-filled inputs ran +19 % on the guide's own GEMM.)"""
import sys
import torch
sys.path.insert(0, "/root/repo")
from spacer_amd import kernels as K
dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


for M, N, Kd in [(8192, 8192, 8192), (5632, 37888, 3584), (5632, 3584, 18944)]:
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for name, mk in [("random N(0,1)", lambda r, c: torch.randn(r, c, device=dev).bfloat16()),
                     ("small ints {0,1}", lambda r, c: torch.randint(0, 2, (r, c), device=dev).bfloat16()),
                     ("zeros", lambda r, c: torch.zeros(r, c, device=dev, dtype=torch.bfloat16))]:
        a, b = mk(M, Kd), mk(N, Kd)
        t = timeit(lambda: K.gemm_nt(a, b, out=out))
        print(f"  {M}x{N}x{Kd}  {name:18s} {t * 1e6:8.1f} us  {2.0 * M * N * Kd / t / 1e12:7.1f} TF/s", flush=True)
