"""GPU parity tests for every C-ABI kernel: HIP result vs the fp32 oracle / a plain torch fp32 restatement of
the same op on the same seeded inputs.  Integer outputs are compared bit-exactly; floating point within the
tolerance written next to each assert (bf16 outputs: a few bf16 ulps of the fp32 result)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import grpo_ref as GR          # noqa: E402
from oracle import qwen2vl_fp32 as O       # noqa: E402
from spacer_amd import kernels as K        # noqa: E402

BF = torch.bfloat16


def rnd(shape, dev, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev)


def assert_close(got, want, atol, rtol, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max():.4g} (tol {atol}+{rtol}*|x|)"


# ----------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 512), (200, 320, 1216), (1040, 3584, 5120),
                                   (17, 1000, 128), (4160, 1280, 1216), (333, 152064 // 8, 256)])
def test_gemm_nt_plain(dev, M, N, K):
    a, b = rnd((M, K), dev, 1), rnd((N, K), dev, 2)
    # asymmetric operands (guide: transpose-detecting check)
    b[:, 0] += 1.0
    got = K_gemm(a, b)
    want = a.float() @ b.float().t()
    assert_close(got, want, atol=0.02 * math.sqrt(K) * 0.1 + 0.02, rtol=1e-2, what=f"gemm {M}x{N}x{K}")


def K_gemm(a, b, **kw):
    return K.gemm_nt(a, b, **kw)


def test_gemm_nt_epilogues(dev):
    M, N, Kd = 300, 640, 320
    a, b = rnd((M, Kd), dev, 3, 0.5), rnd((N, Kd), dev, 4, 0.1)
    bias = rnd((N,), dev, 5)
    base = a.float() @ b.float().t()
    # bias + quick_gelu, bf16 out
    got = K.gemm_nt(a, b, bias=bias, act=K.SPACER_ACT_QUICK_GELU)
    assert_close(got, O.quick_gelu(base + bias.float()), 2e-2, 1e-2, "bias+quick_gelu")
    got = K.gemm_nt(a, b, bias=bias, act=K.SPACER_ACT_GELU_ERF)
    assert_close(got, O.gelu_erf(base + bias.float()), 2e-2, 1e-2, "bias+gelu")
    # fp32 out + fp32 residual (the residual-stream form) and in-place accumulate
    res = rnd((M, N), dev, 6, dtype=torch.float32)
    got = K.gemm_nt(a, b, residual=res, out_dtype=torch.float32)
    assert_close(got, base + res, 2e-3, 1e-3, "f32 residual")
    acc = res.clone()
    K.gemm_nt(a, b, out=acc, residual=acc)
    assert_close(acc, base + res, 2e-3, 1e-3, "f32 accumulate")
    # bf16 residual, strided views
    big = rnd((M, 2 * N), dev, 7)
    got = K.gemm_nt(a, b, residual=big[:, N:])
    assert_close(got, base + big[:, N:].float(), 3e-2, 1e-2, "bf16 residual strided")
    assert K.gemm_nt(a, b, alpha=0.5).float().sub(0.5 * base).abs().max() < 5e-2


@pytest.mark.parametrize("M,N,Kd", [(1300, 1536, 8192), (700, 3584, 8192), (5498, 3584, 8192), (2100, 256, 16384)])
def test_gemm_split_k_tail(dev, M, N, Kd):
    """Shapes whose 256-tiles leave the last round of CUs mostly empty: the tail tiles are cut along K, partial tiles
    meet in the caller's workspace and a second kernel sums them in split order and runs the epilogue.  Must equal the
    unsplit launch up to fp32 summation order and be bit-reproducible."""
    a, b = rnd((M, Kd), dev, 11, 0.5), rnd((N, Kd), dev, 12, 0.1)
    b[:, 0] += 1.0
    bias = rnd((N,), dev, 13)
    res = rnd((M, N), dev, 14, dtype=torch.float32)
    want = a.float() @ b.float().t()
    lib = K._lib.load()
    if K.PLAN.gemm_tile:
        pytest.skip("tile forced by SPACER_GEMM_TILE")
    assert lib.spacer_gemm_tile(M, N, Kd, 1, None) == 256
    for _ in range(3):
        got = K.gemm_nt(a, b, bias=bias, act=K.SPACER_ACT_QUICK_GELU)
        assert_close(got, O.quick_gelu(want + bias.float()), 3e-2, 1e-2, "split-K bias+quick_gelu")
    g1 = K.gemm_nt(a, b, residual=res, out_dtype=torch.float32)
    g2 = K.gemm_nt(a, b, residual=res, out_dtype=torch.float32)
    assert torch.equal(g1, g2), "split-K reduction must be deterministic"
    assert_close(g1, want + res, 3e-3 * math.sqrt(Kd / 256), 1e-3, "split-K f32 residual")
    g0 = K.gemm_nt(a, b, residual=res, out_dtype=torch.float32, split_k=False)
    assert_close(g1, g0, 1e-3, 1e-4, "split vs unsplit")


@pytest.mark.parametrize("M,N,Kd", [(512, 768, 256), (5498, 3584, 4608), (300, 1280, 3840), (77, 1216, 320), (1040, 256, 5120),
                                    (4096, 3584, 1024), (1, 512, 64)])
def test_gemm_trans_b_is_dx(dev, M, N, Kd):
    """dX = dY . W with W read in place ([out, in] = [K, N], trans_b): same bits as the NT kernel on an explicit W^T copy
    (same MFMA sequence, same accumulation order), and close to the fp32 product."""
    dy, w = rnd((M, Kd), dev, 11, 0.5), rnd((Kd, N), dev, 12, 0.1)
    w[0] += 1.0                                                     # asymmetric: a transposed read would show
    got = K.gemm(dy, w, trans_b=True)
    ref = K.gemm_nt(dy, w.t().contiguous())
    from spacer_amd import _lib
    if _lib.load().spacer_gemm_tile(M, N, Kd, 1, None) == 256:           # same tile, same split plan -> the same bits
        assert torch.equal(got, ref), f"trans_b differs from NT-on-transposed: max {float((got.float() - ref.float()).abs().max())}"
    else:                                                          # the NT call ran on the 128 tile: fp32 summation order may differ
        assert_close(got, ref, atol=2e-2, rtol=1e-2, what="dX vs NT (128 tile)")
    assert_close(got, dy.float() @ w.float(), atol=0.05, rtol=2e-2, what=f"dX {M}x{N}x{Kd}")


@pytest.mark.parametrize("T,Nout,Kin", [(256, 512, 768), (5498, 4608, 3584), (4160, 3840, 1280), (333, 256, 1216), (64, 1280, 320),
                                        (1, 8, 8), (70, 264, 72), (2100, 3584, 1536)])
def test_gemm_trans_ab_is_dw(dev, T, Nout, Kin):
    """dW[out, in] += dY[T, out]^T . X[T, in], both operands contraction-major and read in place; T (the contraction length) is
    ragged -- the kernel masks the last K tile.  Compared with the fp32 product and, bit for bit, with the old route (two
    zero-padded transposes + the NT kernel)."""
    dy, x = rnd((T, Nout), dev, 13, 0.5), rnd((T, Kin), dev, 14, 0.5)
    dy[0] += 1.0
    acc0 = rnd((Nout, Kin), dev, 15, dtype=torch.float32)
    got = K.gemm(dy, x, trans_a=True, trans_b=True, out=acc0.clone(), residual=None)
    want = dy.float().t() @ x.float()
    assert_close(got, want, atol=0.02 * math.sqrt(T) * 0.3 + 0.02, rtol=1e-2, what=f"dW {Nout}x{Kin}x{T}")
    # accumulate form (what the engine calls) vs the transpose route
    acc = acc0.clone()
    K.gemm(dy, x, trans_a=True, trans_b=True, out=acc, residual=acc)
    old = acc0.clone()
    dyt, xt = K.transpose_pad(dy), K.transpose_pad(x)
    K.gemm_nt(dyt, xt, out=old, residual=old)
    from spacer_amd import _lib
    if _lib.load().spacer_gemm_tile(Nout, Kin, dyt.shape[1], 1, None) == 256 and dyt.shape[1] // 64 == (T + 63) // 64:
        assert torch.equal(acc, old), f"in-place dW differs from the transpose route: max {float((acc - old).abs().max())}"
    else:
        assert_close(acc, old, atol=1e-3, rtol=1e-4, what="dW vs transpose route (128 tile)")


def test_gemm_trans_rejects_unsupported(dev):
    a, b = rnd((64, 128), dev, 1), rnd((64, 100), dev, 2)
    with pytest.raises(K.SpacerError):
        K.gemm(a, b, trans_a=True, trans_b=True)          # N = 100 is not a multiple of 8
    with pytest.raises(K.SpacerError):
        K.gemm(rnd((128, 100), dev, 3), rnd((100, 64), dev, 4), trans_b=True)   # trans_b alone needs K % 64 == 0


def test_gemm_rejects_bad_k(dev):
    a, b = rnd((64, 96), dev, 1), rnd((64, 96), dev, 2)
    with pytest.raises(K.SpacerError):
        K.gemm_nt(a, b)


@pytest.mark.parametrize("M,N,K", [(64, 3584, 3584), (8, 4608, 3584), (64, 2048, 1536), (33, 1000, 256), (33, 1008, 256), (64, 512, 18944)])
def test_gemm_skinny(dev, M, N, K):
    a, b = rnd((M, K), dev, 1, 0.5), rnd((N, K), dev, 2, 0.05)
    c0 = rnd((M, N), dev, 3, dtype=torch.float32)
    c = c0.clone()
    K_mod = __import__("spacer_amd.kernels", fromlist=["x"])
    K_mod.gemm_skinny_acc(a, b, c)
    assert_close(c, c0 + a.float() @ b.float().t(), 5e-3, 2e-3, "skinny")
    if N % 16 == 0:
        c2 = c0.clone()
        K_mod.gemm_skinny_packed_acc(a, K_mod.pack_weight_frag(b), c2, N)
        assert_close(c2, c0 + a.float() @ b.float().t(), 5e-3, 2e-3, "skinny packed")


@pytest.mark.parametrize("M,I,Kd", [(64, 18944, 3584), (8, 512, 256), (33, 96, 1536), (128, 2048, 1024), (97, 96, 512), (128, 18944, 3584),
                                    (96, 18944, 3584), (40, 17024, 512)])
def test_gemm_skinny_swiglu(dev, M, I, Kd):
    a, w = rnd((M, Kd), dev, 1, 0.5), rnd((2 * I, Kd), dev, 2, 0.05)
    y = K.gemm_skinny_swiglu(a, K.pack_weight_frag_swiglu(w), I)
    gu = a.float() @ w.float().t()
    want = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert_close(y, want, 2e-2, 1e-2, "skinny swiglu")


def test_gemm_skinny_swiglu_tail_balance_leaves_workspace_clean(dev):
    """7B gate|up shape: 592 column groups = one round of 512 + 80; the 80 tail groups run as K-split blocks that meet through
    agent-scope atomics + a ticket.  Repeated launches agree with the fp32 product and with the unbalanced form, and the shared
    workspace is all zero again after every launch."""
    M, I, Kd = 64, 18944, 3584
    a, w = rnd((M, Kd), dev, 31, 0.5), rnd((2 * I, Kd), dev, 32, 0.05)
    wp = K.pack_weight_frag_swiglu(w)
    gu = a.float() @ w.float().t()
    want = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    # default plan at this shape (2368 column fragments on 512 slots): the ONE-ROUND form -- 320 workgroups five fragments wide, 192
    # four; no atomics, so it is bit-reproducible
    outs = [K.gemm_skinny_swiglu(a, wp, I).clone() for _ in range(5)]
    for o in outs:
        assert_close(o, want, 2e-2, 1e-2, "skinny swiglu (one-round form)")
        assert torch.equal(o, outs[0])
    ws = K._SWIGLU_WS[a.device]
    torch.cuda.synchronize()
    assert int(ws.abs().sum()) == 0
    with K.plan(skinny_no_balance=1):
        plain = K.gemm_skinny_swiglu(a, wp, I).clone()
    # fragments owned by ONE wave sum over K in the same order as the plain 64-column launch: bit-identical; the fifth fragment of a wide
    # workgroup (fragment 5 i + 4, i < 320 = output columns 8 (5 i + 4) .. + 7) is summed as four K quarters: fp32 order differs
    fifth = torch.zeros(I, dtype=torch.bool)
    for i in range(2368 - 4 * 512):
        fifth[8 * (5 * i + 4):8 * (5 * i + 4) + 8] = True
    assert torch.equal(outs[0][:, ~fifth.to(dev)], plain[:, ~fifth.to(dev)])
    assert_close(outs[0], plain, 1e-2, 1e-2, "one-round vs plain")
    # a CU budget of 224 (co-running stream beside the decode loop) keeps the old tail-balanced 64-column form: 2368 > 5 x 448
    with K.plan(cus=224):
        bal = K.gemm_skinny_swiglu(a, wp, I).clone()
    assert_close(bal, want, 2e-2, 1e-2, "skinny swiglu (tail-balanced, 448 slots)")
    # fewer rows (cfg4's 8-row decode batch) through the same one-round launch
    o8 = K.gemm_skinny_swiglu(a[:8].contiguous(), wp, I)
    assert torch.equal(o8, outs[0][:8])
    # 65..128 rows (T-GRPO twin rollouts in the same decode batch): two row blocks per weight pass, the same one-round decomposition
    a2 = rnd((128, Kd), dev, 33, 0.5)
    want2 = a2.float() @ w.float().t()
    want2 = torch.nn.functional.silu(want2[:, :I]) * want2[:, I:]
    y2 = K.gemm_skinny_swiglu(a2, wp, I)
    assert_close(y2, want2, 2e-2, 1e-2, "skinny swiglu, 128 rows, one-round form")
    with K.plan(skinny_no_balance=1):
        plain2 = K.gemm_skinny_swiglu(a2, wp, I)
    assert torch.equal(y2[:, ~fifth.to(dev)], plain2[:, ~fifth.to(dev)])
    assert_close(y2, plain2, 1e-2, 1e-2, "one-round vs plain, 128 rows")


@pytest.mark.parametrize("M,N,Kd", [(128, 3584, 3584), (96, 4608, 3584), (65, 512, 18944), (100, 1008, 256)])
def test_gemm_skinny_packed_up_to_128_rows(dev, M, N, Kd):
    """65..128 rows: two 64-row blocks per weight pass (128-wide K slices), packed weights only."""
    a, b = rnd((M, Kd), dev, 1, 0.5), rnd((N, Kd), dev, 2, 0.05)
    c0 = rnd((M, N), dev, 3, dtype=torch.float32)
    c = c0.clone()
    K.gemm_skinny_packed_acc(a, K.pack_weight_frag(b), c, N)
    assert_close(c, c0 + a.float() @ b.float().t(), 5e-3, 2e-3, "skinny packed, M > 64")


@pytest.mark.parametrize("skew", [-1, 1, 7, 17])
@pytest.mark.parametrize("M,N,Kd", [(8, 4608, 3584), (64, 4608, 3584), (33, 1024, 1536), (5, 64, 512)])
def test_gemm_skinny_skewed_k_ranges(dev, M, N, Kd, skew):
    """spacer_plan::skinny_skew forced: equal ranges (-1), an even split derived in the kernel (1 -> alpha 0), the default
    rule's alpha 0.375 (7) and the largest legal skew (17 -> alpha 1, whose first range comes out EMPTY): every K slice is summed
    exactly once -- C against the fp32 product, and for the norm-folded launch the row sums of x^2 as well."""
    a, b = rnd((M, Kd), dev, 1, 0.5), rnd((N, Kd), dev, 2, 0.05)
    bp = K.pack_weight_frag(b)
    c0 = rnd((M, N), dev, 3, dtype=torch.float32)
    want = c0 + a.float() @ b.float().t()
    with K.plan(skinny_skew=skew):
        c = c0.clone()
        K.gemm_skinny_packed_acc(a, bp, c, N)
        assert_close(c, want, 5e-3, 2e-3, f"skinny packed, skew {skew}")
        x32 = rnd((M, Kd), dev, 4, 0.5, torch.float32)
        c2, rowss = c0.clone(), torch.zeros(M, device=dev)
        K.gemm_skinny_packed_normed(x32, bp, c2, rowss, N)
        assert_close(c2, c0 + x32.to(BF).float() @ b.float().t(), 5e-3, 2e-3, f"skinny normed, skew {skew}")
        assert_close(rowss, x32.pow(2).sum(1), 1e-5, 1e-4, f"row sums of x^2, skew {skew}")
    with K.plan(skinny_skew=-1):
        ce = c0.clone()
        K.gemm_skinny_packed_acc(a, bp, ce, N)
    assert_close(c, ce, 1e-4, 1e-4, "skewed vs equal ranges (fp32 summation order only)")


def test_gemm_skinny_skew_out_of_range_is_rejected(dev):
    """alpha = (skew - 1) / 16 > 1 would make the kernel's range boundaries non-monotone (negative starts): refused at the launch."""
    a, b = rnd((8, 512), dev, 1), rnd((64, 512), dev, 2)
    c = torch.zeros(8, 64, device=dev)
    with K.plan(skinny_skew=18):
        with pytest.raises(K.SpacerError, match="skinny_skew"):
            K.gemm_skinny_packed_acc(a, K.pack_weight_frag(b), c, 64)
        with pytest.raises(K.SpacerError, match="skinny_skew"):
            K.gemm_skinny_packed_normed(torch.zeros(8, 512, device=dev), K.pack_weight_frag(b), c, torch.zeros(8, device=dev), 64)


def test_stale_plan_struct_is_rejected(dev):
    """A plan whose struct_bytes is not the library's sizeof(spacer_plan) -- a binding that is a field short, as the round-4
    INTEGRATION.md stub was -- fails with SPACER_EINVAL instead of being over-read."""
    a, b = rnd((256, 64), dev, 1), rnd((256, 64), dev, 2)
    old = K.PLAN.struct_bytes
    try:
        K.PLAN.struct_bytes = old - 4
        with pytest.raises(K.SpacerError, match="struct_bytes"):
            K.gemm_nt(a, b)
    finally:
        K.PLAN.struct_bytes = old
    K.gemm_nt(a, b)


def test_transpose_pad(dev):
    x = rnd((200, 136), dev, 1)
    t = K.transpose_pad(x, 256)
    assert t.shape == (136, 256)
    assert torch.equal(t[:, :200], x.t()) and float(t[:, 200:].float().abs().sum()) == 0.0
    # odd row count, ragged column tile, strided source view (a q|k|v slice), default padding to a multiple of 64
    big = rnd((1403, 3 * 200), dev, 2)
    for view in (big[:, 200:400], big[:, 8:208], big[:77, :64], big[:, 3:131]):     # last one: unaligned -> scalar path
        tt = K.transpose_pad(view)
        R = view.shape[0]
        assert tt.shape == (view.shape[1], (R + 63) // 64 * 64)
        assert torch.equal(tt[:, :R], view.t()) and float(tt[:, R:].float().abs().sum()) == 0.0


# ----------------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("f32", [True, False])
@pytest.mark.parametrize("rows,cols", [(77, 256), (300, 3584), (5, 1280)])
def test_rmsnorm(dev, f32, rows, cols):
    x = rnd((rows, cols), dev, 1, 2.0, torch.float32 if f32 else BF)
    w = rnd((cols,), dev, 2) + 1
    rstd = torch.empty(rows, device=dev)
    y = K.rmsnorm_fwd(x, w, 1e-6, rstd=rstd)
    assert_close(y, O.rms_norm(x.cpu(), w.cpu(), 1e-6), 1e-2, 1e-2, "rmsnorm fwd")
    # backward vs autograd of the oracle
    dy = rnd((rows, cols), dev, 3)
    xr = x.float().cpu().requires_grad_(True); wr = w.float().cpu().requires_grad_(True)
    O.rms_norm(xr, wr, 1e-6).backward(dy.float().cpu())
    dx0 = rnd((rows, cols), dev, 4, dtype=x.dtype)
    dx = dx0.clone(); dw = torch.zeros(cols, device=dev)
    K.rmsnorm_bwd(x, w, dy, rstd, dx, dw, accumulate=True)
    tol = 1e-3 if f32 else 3e-2
    assert_close(dx, dx0.float().cpu() + xr.grad, tol, 1e-2, "rmsnorm dx")
    assert_close(dw, wr.grad, 2e-2 * math.sqrt(rows), 1e-2, "rmsnorm dw")


@pytest.mark.parametrize("f32", [True, False])
def test_layernorm(dev, f32):
    rows, cols = 130, 1280
    x = rnd((rows, cols), dev, 1, 2.0, torch.float32 if f32 else BF) + 0.5
    w = rnd((cols,), dev, 2) + 1; b = rnd((cols,), dev, 5)
    mean = torch.empty(rows, device=dev); rstd = torch.empty(rows, device=dev)
    y = K.layernorm_fwd(x, w, b, 1e-6, mean=mean, rstd=rstd)
    assert_close(y, O.layer_norm(x.cpu(), w.cpu(), b.cpu()), 1e-2, 1e-2, "layernorm fwd")
    dy = rnd((rows, cols), dev, 3)
    xr = x.float().cpu().requires_grad_(True); wr = w.float().cpu().requires_grad_(True); br = b.float().cpu().requires_grad_(True)
    O.layer_norm(xr, wr, br).backward(dy.float().cpu())
    dx = torch.zeros_like(x); dw = torch.zeros(cols, device=dev); db = torch.zeros(cols, device=dev)
    K.layernorm_bwd(x, w, dy, mean, rstd, dx, dw, db, accumulate=False)
    assert_close(dx, xr.grad, 1e-3 if f32 else 3e-2, 1e-2, "layernorm dx")
    assert_close(dw, wr.grad, 0.3, 1e-2, "layernorm dw")
    assert_close(db, br.grad, 0.3, 1e-2, "layernorm db")


# ----------------------------------------------------------------------------------------------- rotary
@pytest.mark.parametrize("D,heads,extra", [(128, 5, 2), (80, 8, 4)])
def test_rope(dev, D, heads, extra):
    T = 37
    x = rnd((T, (heads + extra) * D), dev, 1)
    ang = rnd((T, D // 2), dev, 2, 3.0, torch.float32)
    cos = torch.cat([ang.cos(), ang.cos()], 1).contiguous(); sin = torch.cat([ang.sin(), ang.sin()], 1).contiguous()
    xr = x.float().view(T, heads + extra, D)[:, :heads]
    want = xr * cos[:, None] + O._rot_half(xr.cpu()).to(dev) * sin[:, None]
    y = x.clone()
    K.rope_(y, cos, sin, heads, D)
    assert_close(y.view(T, heads + extra, D)[:, :heads], want, 2e-2, 1e-2, "rope")
    assert torch.equal(y.view(T, heads + extra, D)[:, heads:], x.view(T, heads + extra, D)[:, heads:])
    # inverse is the transpose: <R x, g> == <x, R^T g>
    g = rnd((T, (heads + extra) * D), dev, 3)
    gt = g.clone(); K.rope_(gt, cos, sin, heads, D, inverse=True)
    lhs = (want * g.float().view(T, heads + extra, D)[:, :heads]).sum()
    rhs = (xr * gt.float().view(T, heads + extra, D)[:, :heads]).sum()
    assert abs(float(lhs - rhs)) < 2e-2 * abs(float(lhs)) + 1.0


# ----------------------------------------------------------------------------------------------- attention
def dense_mask(segs, T, causal):
    m = torch.zeros(T, T, dtype=torch.bool)
    for qs, ql, ps, pl in segs:
        for i in range(ql):
            m[qs + i, ps:ps + pl] = True
            m[qs + i, qs:qs + (i + 1 if causal else ql)] = True
    return m


def attn_ref(q, k, v, mask, Hq, Hkv, D, scale):
    T = q.shape[0]
    qh = q.float().view(T, Hq, D).transpose(0, 1)
    kh = k.float().view(T, Hkv, D).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
    vh = v.float().view(T, Hkv, D).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
    s = (qh @ kh.transpose(1, 2)) * scale
    s = s.masked_fill(~mask.to(s.device), float("-inf"))
    return (torch.softmax(s, -1) @ vh).transpose(0, 1).reshape(T, Hq * D)


ATTN_CASES = [
    # (name, D, Hq, Hkv, causal, segments)
    ("vit_frames", 80, 4, 4, False, [(0, 24, 0, 0), (24, 24, 0, 0), (48, 24, 0, 0)]),
    ("vit_520", 80, 2, 2, False, [(0, 520, 0, 0), (520, 520, 0, 0)]),
    ("causal_small", 128, 2, 1, True, [(0, 50, 0, 0)]),
    ("causal_300_gqa", 128, 6, 2, True, [(0, 300, 0, 0)]),
    ("shared_prefix", 128, 4, 2, True, [(0, 45, 0, 0), (45, 6, 0, 45), (51, 6, 0, 45), (57, 6, 0, 45)]),
    ("shared_prefix_big", 128, 7, 1, True, [(0, 200, 0, 0), (200, 130, 0, 200), (330, 130, 0, 200), (460, 70, 0, 200)]),
]


@pytest.mark.parametrize("case", ATTN_CASES, ids=[c[0] for c in ATTN_CASES])
def test_attention_fwd_bwd(dev, case):
    name, D, Hq, Hkv, causal, segs = case
    T = max(s[0] + s[1] for s in segs)
    W = (Hq + 2 * Hkv) * D
    qkv = rnd((T, W), dev, 1, 0.7)
    q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    scale = D ** -0.5
    sd = K.make_segments(segs, dev)
    mq = max(s[1] for s in segs)
    o, lse = K.attn_fwd(q, k, v, sd, mq, Hq, Hkv, D, causal, scale)
    mask = dense_mask(segs, T, causal)
    qr, kr, vr = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    want = attn_ref(qr, kr, vr, mask, Hq, Hkv, D, scale)
    assert_close(o, want.detach(), 2e-2, 2e-2, f"{name} fwd")
    # lse check
    s = (qr.view(T, Hq, D).transpose(0, 1) @ kr.view(T, Hkv, D).repeat_interleave(Hq // Hkv, 1).transpose(0, 1).transpose(1, 2)) * scale
    s = s.masked_fill(~mask.to(dev), float("-inf"))
    assert_close(lse, torch.logsumexp(s, -1).detach(), 2e-2, 1e-3, f"{name} lse")
    # backward
    d_o = rnd((T, Hq * D), dev, 2, 0.5)
    want.backward(d_o.float())
    dqkv = torch.zeros_like(qkv)
    dk32 = torch.zeros(T, Hkv * D, device=dev); dv32 = torch.zeros(T, Hkv * D, device=dev)
    K.attn_bwd(q, k, v, o, d_o, lse, sd, mq, Hq, Hkv, D, causal, scale, dq=dqkv[:, :Hq * D], dk32=dk32, dv32=dv32)
    gs = float(qr.grad.abs().max())
    assert_close(dqkv[:, :Hq * D], qr.grad, 0.03 * gs + 1e-3, 3e-2, f"{name} dq")
    assert_close(dk32, kr.grad, 0.03 * float(kr.grad.abs().max()) + 1e-3, 3e-2, f"{name} dk")
    assert_close(dv32, vr.grad, 0.03 * float(vr.grad.abs().max()) + 1e-3, 3e-2, f"{name} dv")


def test_attention_forced_rescale(dev):
    """Spike one key so the running max jumps at a late tile (rescale branch is exercised)."""
    D, Hq, Hkv, T = 128, 1, 1, 256
    qkv = rnd((T, 3 * D), dev, 5, 0.3)
    qkv[200, D:2 * D] = qkv[255, :D] * 30      # key 200 aligned with query 255
    q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
    sd = K.make_segments([(0, T, 0, 0)], dev)
    o, _ = K.attn_fwd(q, k, v, sd, T, Hq, Hkv, D, True, D ** -0.5)
    want = attn_ref(q, k, v, dense_mask([(0, T, 0, 0)], T, True), Hq, Hkv, D, D ** -0.5)
    assert_close(o, want, 2e-2, 2e-2, "forced rescale")


def test_attention_decode(dev):
    D, Hq, Hkv, B, nP, Pmax, Cmax = 128, 6, 1, 5, 2, 70, 16
    q = rnd((B, Hq * D), dev, 1, 0.7)
    pk, pv = rnd((nP, Pmax, Hkv, D), dev, 2, 0.7), rnd((nP, Pmax, Hkv, D), dev, 3, 0.7)
    tk, tv = rnd((B, Cmax, Hkv, D), dev, 4, 0.7), rnd((B, Cmax, Hkv, D), dev, 5, 0.7)
    plen = torch.tensor([70, 33], dtype=torch.int32, device=dev)
    pof = torch.tensor([0, 0, 1, 1, 1], dtype=torch.int32, device=dev)
    for tl in (0, 9):
        tld = torch.tensor([tl], dtype=torch.int32, device=dev)
        o = K.attn_decode(q, pk, pv, plen, pof, tk, tv, tld, Hq, Hkv, D, D ** -0.5)
        for b in range(B):
            P = int(plen[pof[b]])
            kk = torch.cat([pk[pof[b], :P], tk[b, :tl + 1]]).float()       # [L, Hkv, D]
            vv = torch.cat([pv[pof[b], :P], tv[b, :tl + 1]]).float()
            qq = q[b].float().view(Hq, D)
            kk = kk.repeat_interleave(Hq // Hkv, 1); vv = vv.repeat_interleave(Hq // Hkv, 1)
            s = torch.einsum("hd,lhd->hl", qq, kk) * D ** -0.5
            want = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), vv).reshape(-1)
            assert_close(o[b], want, 2e-2, 2e-2, f"decode attn b={b} tl={tl}")


def test_attention_decode_shared_prefix(dev):
    """Prompt keys scored once per prompt for its Kn rollouts + per-rollout tail == the per-sequence kernel."""
    D, Hq, Hkv, Kn, nP, Pmax, Cmax = 128, 14, 2, 4, 3, 150, 24
    B = nP * Kn
    q = rnd((B, Hq * D), dev, 1, 0.7)
    pk, pv = rnd((nP, Pmax, Hkv, D), dev, 2, 0.7), rnd((nP, Pmax, Hkv, D), dev, 3, 0.7)
    tk, tv = rnd((B, Cmax, Hkv, D), dev, 4, 0.7), rnd((B, Cmax, Hkv, D), dev, 5, 0.7)
    plen = torch.tensor([150, 64, 7], dtype=torch.int32, device=dev)
    pof = (torch.arange(B, device=dev) // Kn).int()
    for tl in (0, 5, 23):
        tld = torch.tensor([tl], dtype=torch.int32, device=dev)
        o = K.attn_decode_shared(q, pk, pv, plen, pof, tk, tv, tld, Kn, Hq, Hkv, D, D ** -0.5)
        for b in range(B):
            P = int(plen[pof[b]])
            kk = torch.cat([pk[pof[b], :P], tk[b, :tl + 1]]).float().repeat_interleave(Hq // Hkv, 1)
            vv = torch.cat([pv[pof[b], :P], tv[b, :tl + 1]]).float().repeat_interleave(Hq // Hkv, 1)
            s = torch.einsum("hd,lhd->hl", q[b].float().view(Hq, D), kk) * D ** -0.5
            want = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), vv).reshape(-1)
            assert_close(o[b], want, 2e-2, 2e-2, f"shared decode attn b={b} tl={tl}")


def test_decode_qkv_projection_with_folded_rmsnorm(dev):
    """gemm_skinny_packed_normed + decode_qkv_finish_normed (RMSNorm folded into the K-split decode projection:
    rstd * (bf16(x) (W diag(w))^T)) against fp32 torch for norm(x) W^T + b with rotary, and against the three-launch path
    (rmsnorm_fwd + gemm_skinny_packed_acc + decode_qkv_finish) to bf16 rounding; the row sums are re-zeroed by the protocol."""
    D, Hq, Hkv, Hd, B, Cmax = 128, 6, 2, 768, 37, 8
    heads = Hq + 2 * Hkv
    x = torch.randn(B, Hd, device=dev, generator=torch.Generator(dev).manual_seed(1)) * 1.7
    w = rnd((heads * D, Hd), dev, 2, 0.04)
    lnw = (1.0 + 0.3 * torch.randn(Hd, device=dev, generator=torch.Generator(dev).manual_seed(3))).to(torch.bfloat16)
    bias = rnd((heads * D,), dev, 4, 0.3)
    eps = 1e-6
    pos_base = torch.arange(B, dtype=torch.int32, device=dev) * 3 + 5
    tld = torch.tensor([2], dtype=torch.int32, device=dev)
    cos, sin = torch.empty(B, D, device=dev), torch.empty(B, D, device=dev)
    K.decode_rope_table(pos_base, tld, 1e6, cos, sin)
    # three launches
    acc_a = torch.zeros(B, heads * D, device=dev)
    h = K.rmsnorm_fwd(x, lnw, eps)
    K.gemm_skinny_packed_acc(h, K.pack_weight_frag(w), acc_a, heads * D)
    q_a = torch.empty(B, Hq * D, device=dev, dtype=torch.bfloat16)
    tk_a, tv_a = torch.zeros(B, Cmax, Hkv, D, device=dev, dtype=torch.bfloat16), torch.zeros(B, Cmax, Hkv, D, device=dev, dtype=torch.bfloat16)
    K.decode_qkv_finish(acc_a, bias, cos, sin, q_a, tk_a, tv_a, tld, Hq, Hkv, D)
    # folded: two launches, twice in a row to exercise the row-sum hand-over (layer i clears layer i+1's sums)
    wn = K.pack_weight_frag((w.float() * lnw.float()[None, :]).to(torch.bfloat16))
    rowss = torch.zeros(2, B, device=dev)
    for it in range(2):
        acc_b = torch.zeros(B, heads * D, device=dev)
        q_b = torch.empty_like(q_a)
        tk_b, tv_b = torch.zeros_like(tk_a), torch.zeros_like(tv_a)
        K.gemm_skinny_packed_normed(x, wn, acc_b, rowss[it], heads * D)
        assert_close(rowss[it], (x.float() ** 2).sum(1), 1e-3 * Hd, 1e-5, "row sums of squares")
        K.decode_qkv_finish_normed(acc_b, bias, cos, sin, q_b, tk_b, tv_b, tld, rowss[it], rowss[1 - it], Hd, eps, Hq, Hkv, D)
        assert float(acc_b.abs().max()) == 0.0 and float(rowss[1 - it].abs().max()) == 0.0
        for a, b, nm in ((q_a, q_b, "q"), (tk_a, tk_b, "k"), (tv_a, tv_b, "v")):
            assert_close(b, a.float(), 3e-2, 2e-2, f"folded norm {nm} vs three launches (it {it})")
        rowss[it].zero_()                     # (in the decode loop the other layer's finishing kernel does this)
    # fp32 reference
    xf = x.float()
    y = (xf * torch.rsqrt((xf ** 2).mean(1, keepdim=True) + eps) * lnw.float()) @ w.float().t() + bias.float()
    yq = y[:, :Hq * D].view(B, Hq, D)
    c, s_ = cos[:, None, :], sin[:, None, :]
    rot = torch.cat([-yq[..., D // 2:], yq[..., :D // 2]], -1)
    assert_close(q_b.view(B, Hq, D), yq * c + rot * s_, 3e-2, 2e-2, "folded norm q vs fp32")
    assert_close(tv_b[:, 2], y[:, (Hq + Hkv) * D:].view(B, Hkv, D), 3e-2, 2e-2, "folded norm v vs fp32")


@pytest.mark.parametrize("M,I,Kd", [(8, 18944, 3584), (16, 18944, 3584), (5, 8960, 1536), (16, 512, 256), (1, 96, 512)])
def test_gemm_skinny_swiglu_small_rows_and_norm_fold(dev, M, I, Kd):
    """<= 16 rows: (i) the SMALL instantiation of the decode gate|up + SwiGLU GEMM (row fragment 0 only) gives the bits of the 64-row
    launch on the same rows; (ii) the norm-folded form -- A = the fp32 stream, rstd from the workgroup's own row sums, W diag(w) in the
    packed weights -- equals rmsnorm + gate|up + SwiGLU to bf16 rounding and the fp32 reference."""
    g = torch.Generator(device="cpu").manual_seed(21)
    x = (torch.randn(M, Kd, generator=g) * 1.3).to(dev)
    w = rnd((2 * I, Kd), dev, 22, 0.05)
    lnw = (1.0 + 0.3 * torch.randn(Kd, generator=g)).to(dev).to(BF)
    eps = 1e-6
    # (i) bf16 A, SMALL vs the 64-row instantiation (rows padded to 17 take the general kernel)
    h = K.rmsnorm_fwd(x, lnw, eps)
    wp = K.pack_weight_frag_swiglu(w)
    y_small = K.gemm_skinny_swiglu(h, wp, I)
    pad = torch.zeros(17, Kd, device=dev, dtype=BF)
    pad[:M] = h
    y_64 = K.gemm_skinny_swiglu(pad, wp, I)[:M]
    assert torch.equal(y_small, y_64)
    # (ii) the fold
    wn = K.pack_weight_frag_swiglu((w.float() * lnw.float()[None, :]).to(BF))
    y_fold = K.gemm_skinny_swiglu_normed(x, wn, I, eps)
    y_fold2 = K.gemm_skinny_swiglu_normed(x, wn, I, eps)
    assert torch.equal(y_fold, y_fold2)
    # exact algebra of the fold on the operands the kernel multiplies: rstd * (bf16(x) . bf16(W diag(w))^T), one bf16 rounding at the end
    xf = x.float()
    rstd = torch.rsqrt((xf ** 2).mean(1, keepdim=True) + eps)
    wnf = (w.float() * lnw.float()[None, :]).to(BF).float()
    gu = (x.to(BF).float() @ wnf.t()) * rstd
    want = torch.nn.functional.silu(gu[:, :I]) * gu[:, I:]
    assert_close(y_fold, want, 2e-3, 1e-2, "norm-folded gate|up + SwiGLU vs its own algebra in fp32")
    # against the unfolded operator (norm -> bf16 h -> GEMM): the two roundings sit at different places, so only statistically close
    hn = xf * rstd * lnw.float()
    gu2 = hn @ w.float().t()
    want2 = torch.nn.functional.silu(gu2[:, :I]) * gu2[:, I:]
    for other, nm in ((want2, "fp32 operator"), (y_small.float(), "norm + gate|up + SwiGLU launches")):
        d = (y_fold.float() - other).abs()
        assert float(d.mean()) <= 6e-3 * float(other.abs().mean()) + 1e-4, (nm, float(d.mean()), float(other.abs().mean()))
        assert float(d.max()) <= 0.05 * float(other.abs().max()), (nm, float(d.max()), float(other.abs().max()))


def test_gemm_swiglu_epilogue_is_the_two_step_path(dev):
    """gate|up GEMM with the SwiGLU in its epilogue == GEMM into gu + swiglu_fwd, bit for bit (ragged M, bias, no gu)."""
    for M, I, Kd, with_bias in ((3000, 2048, 512, False), (4160, 3456, 1280, True), (5498, 1024, 256, False), (1402, 18944, 3584, False)):
        a = rnd((M, Kd), dev, 1, 0.5)
        w = rnd((2 * I, Kd), dev, 2, 0.05)
        b = rnd((2 * I,), dev, 3, 0.2) if with_bias else None
        assert K._lib.load().spacer_gemm_swiglu_fused(M, I, Kd, None) == 1
        gu_ref = K.gemm_nt(a, w, bias=b, split_k=False)
        act_ref = K.swiglu_fwd(gu_ref)
        act, gu = K.gemm_swiglu(a, w, bias=b, keep_gu=True)
        assert torch.equal(gu, gu_ref), f"gu differs: {(gu.float() - gu_ref.float()).abs().max().item()}"
        assert torch.equal(act, act_ref), f"act differs: {(act.float() - act_ref.float()).abs().max().item()}"
        act2, none = K.gemm_swiglu(a, w, bias=b, keep_gu=False)
        assert none is None and torch.equal(act2, act_ref)
    # a shape the 256 tile does not take falls back to the two launches
    a, w = rnd((40, 128), dev, 4, 0.5), rnd((2 * 192, 128), dev, 5, 0.05)
    assert K._lib.load().spacer_gemm_swiglu_fused(40, 192, 128, None) == 0
    act, gu = K.gemm_swiglu(a, w)
    assert torch.equal(act, K.swiglu_fwd(K.gemm_nt(a, w)))


# ----------------------------------------------------------------------------------------------- element-wise
def test_swiglu_act_bias_cast(dev):
    rows, I = 50, 512
    gu = rnd((rows, 2 * I), dev, 1)
    y = K.swiglu_fwd(gu)
    g, u = gu.float()[:, :I], gu.float()[:, I:]
    assert_close(y, torch.nn.functional.silu(g) * u, 1e-2, 1e-2, "swiglu fwd")
    dy = rnd((rows, I), dev, 2)
    gr = gu.float().clone().requires_grad_(True)
    (torch.nn.functional.silu(gr[:, :I]) * gr[:, I:]).backward(dy.float())
    assert_close(K.swiglu_bwd(gu, dy), gr.grad, 2e-2, 1e-2, "swiglu bwd")
    x = rnd((rows, I), dev, 3, 2.0)
    for act, fn in ((K.SPACER_ACT_QUICK_GELU, O.quick_gelu), (K.SPACER_ACT_GELU_ERF, O.gelu_erf)):
        assert_close(K.act_fwd(x, act), fn(x.float()), 1e-2, 1e-2, "act fwd")
        xr = x.float().clone().requires_grad_(True)
        fn(xr).backward(dy.float())
        assert_close(K.act_bwd(x, dy, act), xr.grad, 2e-2, 1e-2, "act bwd")
    db = torch.ones(I, device=dev)
    K.bias_grad_(dy, db)
    assert_close(db, 1 + dy.float().sum(0), 5e-2, 1e-3, "bias grad")
    f = rnd((37, 20), dev, 4, dtype=torch.float32)
    assert torch.equal(K.cast_bf16(f), f.to(BF)) and torch.equal(K.cast_f32(f.to(BF)), f.to(BF).float())
    big = torch.zeros(37, 64, device=dev, dtype=BF)
    K.cast_bf16_strided(f, big[:, 8:28])
    assert torch.equal(big[:, 8:28], f.to(BF)) and float(big[:, :8].float().abs().sum()) == 0


def test_embed(dev):
    V, H, T = 300, 64, 40
    table = rnd((V, H), dev, 1); video = rnd((10, H), dev, 2)
    ids = torch.randint(0, V, (T,), generator=torch.Generator().manual_seed(3)).to(dev)
    vrow = torch.full((T,), -1, dtype=torch.int32); vrow[5:15] = torch.arange(10, dtype=torch.int32); vrow = vrow.to(dev)
    out = K.embed_fwd(ids, table, video, vrow)
    want = table[ids].float(); want[5:15] = video.float()
    assert torch.equal(out, want)
    d_out = rnd((T, H), dev, 4, dtype=torch.float32)
    dt = torch.zeros(V, H, device=dev); dvid = torch.zeros(10, H, device=dev)
    K.embed_bwd(ids, vrow, d_out, dt, dvid)
    wt = torch.zeros(V, H, device=dev); txt = vrow < 0
    wt.index_add_(0, ids[txt], d_out[txt])
    assert_close(dt, wt, 1e-5, 1e-5, "embed bwd table")
    assert torch.equal(dvid, d_out[5:15])


def test_patchify_matches_oracle(dev):
    cfg = O.make_config(hidden=64, layers=1, heads=1, kv_heads=1, intermediate=64, vocab=10, vit_dim=80, vit_depth=1,
                        vit_heads=1, vit_mlp=80)
    for F in (4, 3):
        fr = torch.randint(0, 256, (F, 3, 56, 84), generator=torch.Generator().manual_seed(F), dtype=torch.uint8)
        want, grid = O.patchify_frames(fr, cfg)
        got, g2 = K.patchify(fr.to(dev), kpad=1216)
        assert tuple(g2) == tuple(grid)
        assert_close(got[:, :1176], want, 2e-2, 1e-2, "patchify")
        assert float(got[:, 1176:].float().abs().sum()) == 0


def test_patchify_matches_hf_golden(dev):
    """K1 against HF's own pre-processing output (tests/golden/patchify_hf.npz): bf16 rounding of the normalised pixel only."""
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patchify_hf.npz"))
    for i in range(3):
        got, grid = K.patchify(torch.from_numpy(z[f"frames{i}"]).to(dev), kpad=1216)
        want = torch.from_numpy(z[f"pixel_values{i}"])
        assert tuple(grid) == tuple(int(v) for v in z[f"grid{i}"])
        assert_close(got[:, :1176], want, 1e-2, 4e-3, "patchify vs HF")          # |x| < 2.7, bf16: 2^-9 relative
        assert torch.equal(got[:, :1176].float().cpu(), want.to(BF).float())       # exactly the bf16 rounding of HF's value


# ----------------------------------------------------------------------------------------------- loss
@pytest.mark.parametrize("H,W", [(480, 640), (720, 1280), (448, 448), (60, 80), (100, 100)])
def test_resize_bicubic_antialias_matches_torch(dev, H, W):
    """GPU front end (QU:310-315): spacer_resize_bicubic_aa_u8 against torch's antialiased bicubic on the CPU + torchvision's uint8
    rounding (vision_process.resize_frames), at the target size the reference's smart_resize picks for the source.  Integer
    output: equal except where the fp32 sum lands within round-off of a .5 tie (bounded: < 1e-4 of the pixels, never by more
    than 1 level); the kernel chain resize -> patchify equals the CPU resize -> oracle patchify on all other rows."""
    from spacer_amd.qwen_vl_utils import vision_process as VP
    frames = torch.randint(0, 256, (6, 3, H, W), generator=torch.Generator().manual_seed(H * 7 + W), dtype=torch.uint8)
    _, hw, _ = VP.plan_video({}, 300, 30, H, W)
    want = VP.resize_frames(frames, hw)                                   # CPU, float values on the uint8 grid
    got = VP.resize_frames_gpu(frames.to(dev), hw)
    assert got.dtype == torch.uint8 and tuple(got.shape) == (6, 3) + tuple(hw)
    diff = (got.cpu().float() - want).abs()
    assert float(diff.max()) <= 1.0 and float((diff > 0).float().mean()) < 1e-4, (float(diff.max()), float((diff > 0).float().mean()))
    # frame sampling gather (QU:252) is exact
    idx = torch.tensor(VP.frame_indices(6, 4), dtype=torch.int32)
    if (3 * H * W) % 16 == 0:
        assert torch.equal(K.gather_frames(frames.to(dev), idx.to(dev)).cpu(), frames[idx.long()])


def test_logprob(dev):
    rows, V = 33, 152064 // 16 + 3
    lgp = torch.zeros(rows, (V + 3) // 4 * 4, device=dev)      # row stride must be a multiple of 4 floats
    lgp[:, :V] = rnd((rows, V), dev, 1, 3.0, torch.float32)
    lg = lgp[:, :V]
    tgt = torch.randint(0, V, (rows,), generator=torch.Generator().manual_seed(2)).to(dev)
    lp, lse = K.logprob_fwd(lg, tgt)
    want = torch.log_softmax(lg.double(), -1).gather(1, tgt[:, None]).squeeze(1)
    assert_close(lp, want, 2e-5, 1e-5, "logprob fwd")
    g = rnd((rows,), dev, 3, dtype=torch.float32)
    dlp = torch.zeros(rows, lgp.shape[1], device=dev, dtype=BF)
    dl = K.logprob_bwd(lg, tgt, lse, g, out=dlp[:, :V])
    wantd = (torch.nn.functional.one_hot(tgt, V) - torch.softmax(lg.double(), -1)) * g[:, None]
    assert_close(dl, wantd, 1e-5, 1e-2, "logprob bwd")


def test_grpo_loss_and_mask(dev):
    G, Cc, beta = 8, 70, 0.04
    gen = torch.Generator().manual_seed(0)
    lp = -torch.rand(G, Cc, generator=gen) * 3
    ref = lp + torch.randn(G, Cc, generator=gen) * 0.5
    ref[0, 0] = lp[0, 0] + 15; ref[0, 1] = lp[0, 1] - 15          # clamp edges
    ids = torch.randint(5, 100, (G, Cc), generator=gen); ids[1, 10] = 7; ids[1, 20] = 7; ids[2, 0] = 7; ids[3, Cc - 1] = 7
    mask = GR.completion_mask(ids, 7)
    m2, lens = K.completion_mask(ids.to(dev), 7)
    assert torch.equal(m2.cpu(), mask) and torch.equal(lens.cpu().long(), mask.sum(1))
    adv = torch.randn(G, generator=gen)
    loss_o, grad_o = GR.grpo_loss_and_grad(lp, ref, adv, mask, beta)
    loss, kl, dlp = K.grpo_loss(lp.to(dev), ref.to(dev), adv.to(dev), m2, beta)
    assert_close(loss, loss_o.reshape(1), 1e-5, 1e-5, "grpo loss")
    assert_close(kl, GR.kl_metric(lp, ref, mask).reshape(1), 1e-5, 1e-5, "kl metric")
    assert_close(dlp, grad_o, 1e-7, 1e-4, "grpo dlogp")


def test_adamw_and_sumsq(dev):
    n = 10007
    p0 = rnd((n,), dev, 1, dtype=torch.float32); g = rnd((n,), dev, 2, 3.0, dtype=torch.float32)
    ref_p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([ref_p], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    p = p0.clone(); m = torch.zeros_like(p); v = torch.zeros_like(p); sh = torch.empty(n, device=dev, dtype=BF)
    for step in (1, 2, 3):
        acc = torch.zeros(1, device=dev)
        K.sumsq_(g, acc)
        assert_close(acc, (g.double() ** 2).sum().reshape(1), 1e-2, 1e-5, "sumsq")
        ref_p.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([ref_p], 5.0)
        opt.step()
        K.adamw_step_(p, sh, m, v, g, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=step,
                      sumsq=acc, max_norm=5.0)
        assert_close(p, ref_p.data, 1e-6, 1e-5, f"adamw step {step}")
        assert torch.equal(sh, p.to(BF))


# ----------------------------------------------------------------------------------------------- sampler
def test_sampler_support_and_distribution(dev):
    V, B = 5000, 64
    g = torch.Generator().manual_seed(0)
    base = torch.randn(V, generator=g) * 2
    logits = base.repeat(B, 1).to(dev).contiguous()
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    # exact support: HF chain = top_k 50 then top_p 0.95 (remove ascending cumsum <= 0.05)
    tk = torch.topk(base, 50)
    pr = torch.softmax(tk.values.double(), 0)
    asc = torch.flip(pr, [0]).cumsum(0)
    removed = int((asc <= 1 - 0.95).sum())
    support = set(tk.indices[:50 - removed].tolist())
    probs = (pr[:50 - removed] / pr[:50 - removed].sum()).numpy()
    counts = {}
    n_draw = 0
    for s in range(60):
        step.fill_(s)
        ids = K.sample_top_p(logits, step, top_k=50, top_p=0.95, seed=1234)
        for t in ids.tolist():
            assert t in support, f"token {t} outside the nucleus"
            counts[t] = counts.get(t, 0) + 1
        n_draw += B
    # chi-square on the 8 most likely tokens + rest
    import numpy as np
    order = tk.indices[:50 - removed].tolist()
    obs = np.array([counts.get(t, 0) for t in order[:8]] + [sum(counts.get(t, 0) for t in order[8:])], dtype=float)
    exp = np.array(list(probs[:8]) + [probs[8:].sum()]) * n_draw
    chi2 = float(((obs - exp) ** 2 / exp).sum())
    assert chi2 < 40.0, (chi2, obs, exp)          # 8 dof, p ~ 1e-5
    # determinism for a fixed (seed, step); eos suppression and finished rows
    step.fill_(3)
    a = K.sample_top_p(logits, step, seed=7); b = K.sample_top_p(logits, step, seed=7)
    assert torch.equal(a, b)
    top = int(base.argmax())
    fin = torch.zeros(B, dtype=torch.int32, device=dev); fin[5] = 1
    ids = K.sample_top_p(logits, step, top_k=1, top_p=1.0, eos_id=top, pad_id=4242, finished=fin)
    assert int(ids[5]) == 4242 and all(int(t) == top for i, t in enumerate(ids.tolist()) if i != 5)
    assert int(fin.sum()) == B                      # every live row drew eos
    ids = K.sample_top_p(logits, step, top_k=1, top_p=1.0, eos_id=top, suppress_eos=True)
    assert all(int(t) != top for t in ids.tolist())


def test_skinny_gemm_store_form(dev):
    """lm_head form of the packed decode GEMM: C = A.W^T by plain stores (stale C contents must not leak in)."""
    M, N, Kd = 37, 448 * 64 + 128, 512
    a, w = rnd((M, Kd), dev, 1, 0.5), rnd((N, Kd), dev, 2, 0.05)
    wp = K.pack_weight_frag(w)
    c = torch.full((M, N), 7.0, device=dev)
    K.gemm_skinny_packed_store(a, wp, c, N)
    ref = torch.zeros(M, N, device=dev)
    K.gemm_skinny_packed_acc(a, wp, ref, N)
    assert torch.equal(c, ref)
    with pytest.raises(K.SpacerError):                       # narrow N splits K across workgroups: no store form
        K.gemm_skinny_packed_store(a, K.pack_weight_frag(w[:1024]), torch.zeros(M, 1024, device=dev), 1024)


def test_sampler_mass_ties_take_the_exact_select_path(dev):
    """More than CAP (1024) logits tied at the top: the candidate bound overflows and the exact 4-pass radix select over the whole
    row runs (the path whose bin walk is done by one wave).  Every draw must be one of the tied maxima; and with a strict top-k
    below the tie level the support is exactly those k tokens."""
    V, B = 6000, 16
    g = torch.Generator().manual_seed(3)
    base = torch.randn(V, generator=g)
    tied = torch.randperm(V, generator=g)[:2000]
    base[tied] = 5.0
    logits = base.repeat(B, 1).to(dev).contiguous()
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    tied_set = set(tied.tolist())
    for s in range(4):
        step.fill_(s)
        ids = K.sample_top_p(logits, step, top_k=50, top_p=0.95, seed=5)
        assert all(t in tied_set for t in ids.tolist())
    base2 = base.clone()
    hot = torch.randperm(V, generator=g)[:7]
    base2[hot] = torch.tensor([9.0, 8.5, 8.0, 7.5, 7.0, 6.5, 6.0])
    logits2 = base2.repeat(B, 1).to(dev).contiguous()
    seen = set()
    for s in range(40):
        step.fill_(s)
        seen.update(K.sample_top_p(logits2, step, top_k=5, top_p=1.0, seed=9).tolist())
    assert seen <= set(hot[:5].tolist()) and len(seen) >= 3


@pytest.mark.parametrize("B", [1, 8, 12, 64, 100])
def test_sampler_wide_form_draws_the_same_tokens(dev, B):
    """Round 6: the sampler's wide form (several workgroups per row for the two passes over the logits: partial maxima -> bound -> candidate
    gather -> the same sort / top-k trim / nucleus / draw) must return the SAME token as the one-workgroup-per-row form for the same logits
    and Philox counter: Qwen2-VL's vocabulary (152 064, and a ragged 152 067 in a padded buffer), per-row different logits, several steps,
    top_k 1 / 50 / 1024, temperature, EOS suppression, finished rows; a row with 2000 tied maxima overflows the candidate list and takes the
    single-workgroup fallback inside the final kernel."""
    for V in (152064, 152067):
        Vp = (V + 3) // 4 * 4
        g = torch.Generator().manual_seed(B + V)
        buf = torch.zeros(B, Vp, device=dev)
        buf[:, :V] = (torch.randn(B, V, generator=g) * 2.5).to(dev)
        if B >= 8:
            buf[3, torch.randperm(V, generator=g)[:2000].to(dev)] = 11.0       # mass tie: > CAP candidates in this row
        logits = buf[:, :V]
        ws = K.sample_workspace(B, V, dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev)
        fin0 = torch.zeros(B, dtype=torch.int32, device=dev)
        if B > 5:
            fin0[5] = 1
        for s, (tk, tp, temp, sup) in enumerate(((50, 0.95, 1.0, False), (1, 1.0, 1.0, True), (1024, 0.9, 0.7, False), (50, 1.0, 1.3, True), (7, 0.5, 1.0, False))):
            step.fill_(s)
            eos = int(logits[0].argmax())
            fa, fb = fin0.clone(), fin0.clone()
            a = K.sample_top_p(logits, step, top_k=tk, top_p=tp, temperature=temp, seed=11, eos_id=eos, pad_id=3, suppress_eos=sup, finished=fa)
            b = K.sample_top_p(logits, step, top_k=tk, top_p=tp, temperature=temp, seed=11, eos_id=eos, pad_id=3, suppress_eos=sup, finished=fb,
                               workspace=ws)
            assert torch.equal(a, b), (V, s, (a != b).nonzero().flatten().tolist())
            assert torch.equal(fa, fb)
        # the decode-loop form (token also written into the [B, C] matrix at the device-side step)
        out_a = torch.zeros(B, 4, dtype=torch.int64, device=dev); out_b = torch.zeros_like(out_a)
        ia, ib = torch.empty(B, dtype=torch.int64, device=dev), torch.empty(B, dtype=torch.int64, device=dev)
        step.fill_(1)
        K.sample_top_p_step(logits, step, 1, out_a, seed=5, out_ids=ia)
        K.sample_top_p_step(logits, step, 1, out_b, seed=5, out_ids=ib, workspace=ws)
        assert torch.equal(out_a, out_b) and torch.equal(ia, ib) and bool((out_a[:, 2] == ia).all())
