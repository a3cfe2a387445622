"""Yardstick, not product: the step's GEMM shapes through torch.matmul (hipBLASLt / rocBLAS as shipped in the image) beside this
library's 256-tile kernel, in one process on one box, alternating, HIP events around runs of launches.  Says how much of the gap to
the 2.5 PFLOP/s dense bf16 peak a vendor-tuned kernel closes on the same silicon at the same clocks.

    python scripts/probes/gemm_vs_library.py"""
import statistics

import torch

from spacer_amd import kernels as K

REPS, LAUNCHES = 7, 6


def main():
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)      # noqa: E731
    T, H, I, QKV = 10996, 3584, 18944, 4608
    x, xi, dgu = rnd(T, H), rnd(T, I), rnd(T, 2 * I)
    w_gu, w_qkv, w_o, w_dn = rnd(2 * I, H), rnd(QKV, H), rnd(H, H), rnd(H, I)
    cases = [
        ("gate|up NT (T x 37888 x 3584)", 2.0 * T * 2 * I * H, lambda: K.gemm_nt(x, w_gu), lambda: torch.matmul(x, w_gu.t())),
        ("q|k|v NT (T x 4608 x 3584)", 2.0 * T * QKV * H, lambda: K.gemm_nt(x, w_qkv), lambda: torch.matmul(x, w_qkv.t())),
        ("o NT (T x 3584 x 3584)", 2.0 * T * H * H, lambda: K.gemm_nt(x, w_o), lambda: torch.matmul(x, w_o.t())),
        ("down NT (T x 3584 x 18944)", 2.0 * T * H * I, lambda: K.gemm_nt(xi, w_dn), lambda: torch.matmul(xi, w_dn.t())),
        ("dX of gate|up NN (T x 3584 x 37888)", 2.0 * T * H * 2 * I, lambda: K.gemm(dgu, w_gu, trans_b=True), lambda: torch.matmul(dgu, w_gu)),
        ("dW of gate|up TN (37888 x 3584 x T)", 2.0 * T * H * 2 * I, lambda: K.gemm(dgu, x, trans_a=True, trans_b=True), lambda: torch.matmul(dgu.t(), x)),
    ]
    print(f"| shape (T = {T}, bf16 in / bf16 out) | this library TF/s (median of {REPS}, min-max) | torch.matmul TF/s |")
    print("|---|---:|---:|")
    for name, flops, ours, lib in cases:
        res = {"ours": [], "lib": []}
        for fn in (ours, lib):
            fn()
        torch.cuda.synchronize()
        for _ in range(REPS):
            for key, fn in (("ours", ours), ("lib", lib)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(LAUNCHES):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[key].append(flops * LAUNCHES / (e0.elapsed_time(e1) * 1e-3) / 1e12)
        cell = lambda v: f"{statistics.median(v):.0f} ({min(v):.0f}-{max(v):.0f})"      # noqa: E731
        print(f"| {name} | {cell(res['ours'])} | {cell(res['lib'])} |")


if __name__ == "__main__":
    main()
