"""Per-kernel microbenchmarks on the shapes of the cfg3 step (run on the GPU box):
    python scripts/bench_kernels.py [gemm|skinny|decode|sampler|all]
Prints achieved TFLOP/s or TB/s per shape with HIP-event timing (median of interleaved rounds)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacer_amd import kernels as K  # noqa: E402

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


def rnd(*s, scale=1.0):
    return (torch.randn(*s, device=dev) * scale).to(BF)


def bench_gemm():
    print("== gemm_bf16_nt (M,N,K) -> TFLOP/s")
    for M, N, K_ in [(5496, 4608, 3584), (5496, 3584, 3584), (5496, 37888, 3584), (5496, 3584, 18944), (4096, 152064, 3584),
                     (3584, 18944, 5504), (37888, 3584, 5504), (4160, 3840, 1280), (4160, 5120, 1280), (4160, 1280, 5120),
                     (1400, 37888, 3584), (8192, 8192, 8192), (4096, 4096, 4096)]:
        a, b = rnd(M, K_), rnd(N, K_)
        out = torch.empty(M, N, device=dev, dtype=BF)
        t = timeit(lambda: K.gemm_nt(a, b, out=out), iters=10)
        print(f"  {M:6d} {N:6d} {K_:6d}: {2 * M * N * K_ / t / 1e12:8.1f} TF/s  {t * 1e6:9.1f} us")


def bench_skinny():
    print("== gemm_skinny (M,N,K) -> TB/s of weights")
    for M, N, K_ in [(64, 4608, 3584), (64, 3584, 3584), (64, 37888, 3584), (64, 3584, 18944), (64, 152064, 3584), (8, 37888, 3584)]:
        ws = [rnd(N, K_, scale=0.02) for _ in range(max(1, int(3e9 // (N * K_ * 2))))]   # rotate weights: defeat the 256 MB L3
        a = rnd(M, K_)
        c = torch.zeros(M, N, device=dev)
        i = [0]

        def f():
            K.gemm_skinny_acc(a, ws[i[0] % len(ws)], c); i[0] += 1
        t = timeit(f, iters=30)
        wp = [K.pack_weight_frag(w) for w in ws]
        j = [0]

        def fp():
            K.gemm_skinny_packed_acc(a, wp[j[0] % len(wp)], c, N); j[0] += 1
        tp = timeit(fp, iters=30)
        print(f"  {M:3d} {N:6d} {K_:6d}: row-major {N * K_ * 2 / t / 1e12:5.2f} TB/s {t * 1e6:7.1f} us | packed {N * K_ * 2 / tp / 1e12:5.2f} TB/s {tp * 1e6:7.1f} us")


def bench_decode_attn():
    print("== attn_decode B=64 Hq=28 Hkv=4 D=128")
    B, Hq, Hkv, D, nP, P, C = 64, 28, 4, 128, 8, 1402, 512
    q = rnd(B, Hq * D)
    pk, pv, tk, tv = rnd(nP, P, Hkv, D), rnd(nP, P, Hkv, D), rnd(B, C, Hkv, D), rnd(B, C, Hkv, D)
    plen = torch.full((nP,), P, dtype=torch.int32, device=dev)
    pof = (torch.arange(B, device=dev) // 8).int()
    for tl in (0, 255, 511):
        tld = torch.tensor([tl], dtype=torch.int32, device=dev)
        o = torch.empty(B, Hq * D, device=dev, dtype=BF)
        t = timeit(lambda: K.attn_decode(q, pk, pv, plen, pof, tk, tv, tld, Hq, Hkv, D, D ** -0.5, out=o))
        kv = (nP * P + B * (tl + 1)) * Hkv * D * 2 * 2
        t2 = timeit(lambda: K.attn_decode_shared(q, pk, pv, plen, pof, tk, tv, tld, 8, Hq, Hkv, D, D ** -0.5, out=o))
        print(f"  tail {tl:4d}: per-seq {t * 1e6:7.1f} us | shared-prefix {t2 * 1e6:7.1f} us   unique KV {kv / 1e6:7.1f} MB -> {kv / t2 / 1e12:5.2f} TB/s")


def bench_attn():
    print("== prefill/scoring attention, cfg3 group layout: prompt 1402 + 8 x 512 rollouts sharing the prompt keys")
    P, C, Kn, Hq, Hkv, D = 1402, 512, 8, 28, 4, 128
    T = P + Kn * C
    qkv = rnd(T, (Hq + 2 * Hkv) * D)
    q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    segs = K.make_segments([(0, P, 0, 0)] + [(P + i * C, C, 0, P) for i in range(Kn)], dev)
    pairs = P * (P + 1) / 2 + Kn * (C * P + C * (C + 1) / 2)          # visible (q, k) pairs per head
    flops = 4.0 * pairs * D * Hq
    o, lse = K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5)
    t = timeit(lambda: K.attn_fwd(q, k, v, segs, P, Hq, Hkv, D, True, D ** -0.5, out=o, lse=lse), iters=20)
    print(f"  fwd  D=128: {t * 1e6:8.1f} us  {flops / t / 1e12:7.1f} TF/s")
    d_o = rnd(T, Hq * D)
    dqkv = torch.empty_like(qkv)
    dk32 = torch.zeros(T, Hkv * D, device=dev); dv32 = torch.zeros(T, Hkv * D, device=dev)
    t = timeit(lambda: K.attn_bwd(q, k, v, o, d_o, lse, segs, P, Hq, Hkv, D, True, D ** -0.5, dq=dqkv[:, :Hq * D], dk32=dk32, dv32=dv32),
               iters=10)
    print(f"  bwd  D=128: {t * 1e6:8.1f} us  {2.5 * flops / t / 1e12:7.1f} TF/s (2.5x forward flops)")
    # ViT: 8 temporal grids of 520 patches, 16 heads x 80, non-causal
    Tv, Hv, Dv = 4160, 16, 80
    qkv = rnd(Tv, 3 * Hv * Dv)
    segs = K.make_segments([(i * 520, 520, 0, 0) for i in range(8)], dev)
    fl = 4.0 * 8 * 520 * 520 * Dv * Hv
    t = timeit(lambda: K.attn_fwd(qkv[:, :Hv * Dv], qkv[:, Hv * Dv:2 * Hv * Dv], qkv[:, 2 * Hv * Dv:], segs, 520, Hv, Hv, Dv, False, Dv ** -0.5),
               iters=20)
    print(f"  fwd  D=80 (ViT): {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TF/s")


def bench_sampler():
    print("== sampler B=64 V=152064")
    lg = torch.randn(64, 152064, device=dev)
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty(64, dtype=torch.int64, device=dev)
    t = timeit(lambda: K.sample_top_p(lg, step, out_ids=out))
    print(f"  {t * 1e6:8.1f} us")


def bench_misc():
    print("== transpose / norms")
    x = rnd(5496, 18944)
    t = timeit(lambda: K.transpose_pad(x))
    print(f"  transpose 5496x18944: {t * 1e6:8.1f} us  {x.numel() * 4 / t / 1e12:5.2f} TB/s")
    x32 = torch.randn(5496, 3584, device=dev); w = rnd(3584)
    t = timeit(lambda: K.rmsnorm_fwd(x32, w, 1e-6))
    print(f"  rmsnorm 5496x3584 f32->bf16: {t * 1e6:8.1f} us  {x32.numel() * 6 / t / 1e12:5.2f} TB/s")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("gemm", "all"):
        bench_gemm()
    if what in ("skinny", "all"):
        bench_skinny()
    if what in ("decode", "all"):
        bench_decode_attn()
    if what in ("attn", "all"):
        bench_attn()
    if what in ("sampler", "all"):
        bench_sampler()
    if what in ("misc", "all"):
        bench_misc()
