"""Randomised parity sweep of the hot kernels against fp32 torch on the GPU:  python scripts/fuzz_kernels.py [seconds] [seed].
The fixed-shape parity tests live in tests/; this sweep draws ragged shapes / segment layouts to look for corner cases
(ragged last tiles, K-split tails, ragged contraction lengths of the in-place dW GEMM, tails crossing key tiles, windows of odd
sizes).  A mismatch raises AssertionError; tests/test_fuzz_gpu.py runs a seeded 45-second slice of it under `-m gpu`."""
import math
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from spacer_amd import kernels as K  # noqa: E402
from test_kernels_gpu import attn_ref, dense_mask  # noqa: E402

dev = torch.device("cuda:0")         # the module is imported on the GPU box only (script, or tests/test_fuzz_gpu.py)
BF = torch.bfloat16


def rnd(shape, scale=1.0, dtype=BF):
    return (torch.randn(*shape, device=dev) * scale).to(dtype)


def close(got, want, atol, rtol, what):
    err = (got.float() - want.float()).abs()
    tol = atol + rtol * want.float().abs()
    if bool((err > tol).any()):
        raise AssertionError(f"MISMATCH {what}: max err {float(err.max()):.4g}, {int((err > tol).sum())}/{err.numel()} off")


def fuzz_gemm(r):
    M = r.choice([r.randint(1, 700), r.randint(700, 6000)])
    N = r.choice([r.randint(1, 600) * 8, r.randint(1, 40) * 256, r.randint(8, 80) * 128])
    Kd = r.randint(1, 40) * 64
    a, b = rnd((M, Kd), 0.5), rnd((N, Kd), 0.1)
    want = a.float() @ b.float().t()
    mode = r.randint(0, 3)
    tol = 0.02 * math.sqrt(Kd) * 0.05 + 0.02
    if mode == 0:
        close(K.gemm_nt(a, b), want, tol, 1e-2, f"gemm {M}x{N}x{Kd}")
    elif mode == 1:
        bias = rnd((N,))
        close(K.gemm_nt(a, b, bias=bias, act=K.SPACER_ACT_SILU), torch.nn.functional.silu(want + bias.float()), tol, 1e-2,
              f"gemm bias+silu {M}x{N}x{Kd}")
    elif mode == 2:
        res = rnd((M, N), dtype=torch.float32)
        close(K.gemm_nt(a, b, residual=res, out_dtype=torch.float32), want + res, tol, 1e-2, f"gemm f32+res {M}x{N}x{Kd}")
    else:
        c = rnd((M, N), dtype=torch.float32)
        c0 = c.clone()
        K.gemm_nt(a, b, out=c, residual=c, out_dtype=torch.float32, split_k=bool(r.randint(0, 1)))
        close(c, want + c0, tol, 1e-2, f"gemm accumulate {M}x{N}x{Kd}")


def fuzz_gemm_trans(r):
    """The backward GEMMs with operands read in place: dX = dY . W (trans_b) and dW += dY^T . X (trans_a + trans_b, ragged K)."""
    if r.randint(0, 1):
        M, N, Kd = r.choice([r.randint(1, 700), r.randint(700, 6000)]), r.randint(1, 500) * 8, r.randint(1, 40) * 64
        dy, w = rnd((M, Kd), 0.5), rnd((Kd, N), 0.1)
        close(K.gemm(dy, w, trans_b=True), dy.float() @ w.float(), 0.02 * math.sqrt(Kd) * 0.05 + 0.02, 1e-2, f"gemm dX {M}x{N}x{Kd}")
    else:
        T, Nout, Kin = r.choice([r.randint(1, 300), r.randint(300, 6000)]), r.randint(1, 500) * 8, r.randint(1, 500) * 8
        dy, x = rnd((T, Nout), 0.5), rnd((T, Kin), 0.5)
        acc = rnd((Nout, Kin), dtype=torch.float32)
        want = acc + dy.float().t() @ x.float()
        K.gemm(dy, x, trans_a=True, trans_b=True, out=acc, residual=acc)
        close(acc, want, 0.02 * math.sqrt(T) * 0.3 + 0.02, 1e-2, f"gemm dW {Nout}x{Kin}x{T}")


def fuzz_resize(r):
    """GPU bicubic-antialias resize vs torch's operator + uint8 rounding (at most one level off, on < 1e-3 of the pixels)."""
    from spacer_amd.qwen_vl_utils import vision_process as VP
    H, W = r.randint(20, 300), r.randint(20, 400)
    h, w = r.randint(1, 12) * 28, r.randint(1, 14) * 28
    fr = torch.randint(0, 256, (r.randint(1, 3), 3, H, W), dtype=torch.uint8)
    got = VP.resize_frames_gpu(fr.to(dev), (h, w)).cpu().float()
    d = (got - VP.resize_frames(fr, (h, w))).abs()
    if float(d.max()) > 1.0 or float((d > 0).float().mean()) > 1e-3:
        raise AssertionError(f"MISMATCH resize {H}x{W}->{h}x{w}: max {float(d.max())}, frac {float((d > 0).float().mean()):.2e}")


def fuzz_swiglu(r):
    M, I, Kd = r.randint(1, 6000), r.randint(1, 40) * 128, r.randint(1, 30) * 64
    a, w = rnd((M, Kd), 0.5), rnd((2 * I, Kd), 0.08)
    bias = rnd((2 * I,), 0.2) if r.randint(0, 1) else None
    act, gu = K.gemm_swiglu(a, w, bias=bias, keep_gu=True)
    gu_ref = K.gemm_nt(a, w, bias=bias, split_k=False)
    if not torch.equal(gu, gu_ref) or not torch.equal(act, K.swiglu_fwd(gu_ref)):
        raise AssertionError(f"MISMATCH gemm_swiglu {M}x{I}x{Kd} (fused={K._lib.load().spacer_gemm_swiglu_fused(M, I, Kd, None)})")


def fuzz_attention(r):
    D, Hq, Hkv = r.choice([(128, 4, 2), (128, 7, 1), (80, 3, 3)])
    causal = D == 128
    segs, at = [], 0
    if causal and r.randint(0, 1):
        P = r.randint(1, 300)
        segs.append((0, P, 0, 0)); at = P
        for _ in range(r.randint(1, 5)):
            L = r.randint(1, 200)
            segs.append((at, L, 0, P)); at += L
    else:
        for _ in range(r.randint(1, 5)):
            L = r.randint(1, 330)
            segs.append((at, L, 0, 0)); at += L
    T = at
    qkv = rnd((T, (Hq + 2 * Hkv) * D), 0.7)
    q, k, v = qkv[:, :Hq * D], qkv[:, Hq * D:(Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    sd = K.make_segments(segs, dev)
    mq = max(s[1] for s in segs)
    o, lse = K.attn_fwd(q, k, v, sd, mq, Hq, Hkv, D, causal, D ** -0.5)
    mask = dense_mask(segs, T, causal)
    qr, kr, vr = (t.float().detach().clone().requires_grad_(True) for t in (q, k, v))
    want = attn_ref(qr, kr, vr, mask, Hq, Hkv, D, D ** -0.5)
    close(o, want, 2e-2, 2e-2, f"attn fwd {segs}")
    do = rnd((T, Hq * D), 0.5)
    want.backward(do.float())
    dqkv = torch.zeros_like(qkv)
    dq = dqkv[:, :Hq * D]
    dk, dv = torch.zeros(T, Hkv * D, device=dev), torch.zeros(T, Hkv * D, device=dev)
    K.attn_bwd(q, k, v, o, do, lse, sd, mq, Hq, Hkv, D, causal, D ** -0.5, dq=dq, dk32=dk, dv32=dv)
    close(dq, qr.grad, 3e-2, 3e-2, f"attn dq {segs}")
    close(dk, kr.grad, 3e-2 * math.sqrt(max(1, len(segs))), 3e-2, f"attn dk {segs}")
    close(dv, vr.grad, 3e-2 * math.sqrt(max(1, len(segs))), 3e-2, f"attn dv {segs}")


def fuzz_decode_attention(r):
    D, Hkv = 128, r.choice([1, 2, 4])
    rep = r.choice([1, 2, 4, 7])
    Hq = Hkv * rep
    Kn = r.randint(2, max(2, 64 // rep)) if rep * 2 <= 64 else 2
    Kn = min(Kn, 64 // rep)
    nP = r.randint(1, 4)
    B = nP * Kn
    Pmax, Cmax = r.randint(1, 400), r.randint(1, 300)
    q = rnd((B, Hq * D), 0.7)
    pk, pv = rnd((nP, Pmax, Hkv, D), 0.7), rnd((nP, Pmax, Hkv, D), 0.7)
    tk, tv = rnd((B, Cmax, Hkv, D), 0.7), rnd((B, Cmax, Hkv, D), 0.7)
    plen = torch.tensor([r.randint(1, Pmax) for _ in range(nP)], dtype=torch.int32, device=dev)
    pof = (torch.arange(B, device=dev) // Kn).int()
    tl = r.randint(0, Cmax - 1)
    tld = torch.tensor([tl], dtype=torch.int32, device=dev)
    o = K.attn_decode_shared(q, pk, pv, plen, pof, tk, tv, tld, Kn, Hq, Hkv, D, D ** -0.5)
    o2 = K.attn_decode(q, pk, pv, plen, pof, tk, tv, tld, Hq, Hkv, D, D ** -0.5)
    for b in range(B):
        P = int(plen[pof[b]])
        kk = torch.cat([pk[pof[b], :P], tk[b, :tl + 1]]).float().repeat_interleave(rep, 1)
        vv = torch.cat([pv[pof[b], :P], tv[b, :tl + 1]]).float().repeat_interleave(rep, 1)
        s = torch.einsum("hd,lhd->hl", q[b].float().view(Hq, D), kk) * D ** -0.5
        want = torch.einsum("hl,lhd->hd", torch.softmax(s, -1), vv).reshape(-1)
        close(o[b], want, 2e-2, 2e-2, f"decode attn shared nP={nP} Kn={Kn} rep={rep} Hkv={Hkv} P={P} tl={tl}")
        close(o2[b], want, 2e-2, 2e-2, f"decode attn plain nP={nP} Kn={Kn} rep={rep} Hkv={Hkv} P={P} tl={tl}")


def fuzz_skinny(r):
    M = r.randint(1, 128)
    N, Kd = r.randint(1, 300) * 16, r.randint(1, 24) * 256
    a, b = rnd((M, Kd), 0.5), rnd((N, Kd), 0.05)
    c0 = rnd((M, N), dtype=torch.float32)
    c = c0.clone()
    K.gemm_skinny_packed_acc(a, K.pack_weight_frag(b), c, N)
    close(c, c0 + a.float() @ b.float().t(), 5e-3, 3e-3, f"skinny packed {M}x{N}x{Kd}")
    if M <= 128 and N % 64 == 0:
        I = N // 2
        y = K.gemm_skinny_swiglu(a, K.pack_weight_frag_swiglu(b), I)
        gu = a.float() @ b.float().t()
        close(y, torch.nn.functional.silu(gu[:, :I]) * gu[:, I:], 2e-2, 1e-2, f"skinny swiglu {M}x{I}x{Kd}")


def fuzz_skinny_normed(r):
    """RMSNorm folded into the K-split decode projection: c += bf16(x) (W diag(w))^T with the row sums of x^2 beside it."""
    M = r.randint(1, 64)
    N, Kd = r.randint(1, 300) * 16, r.randint(1, 24) * 256
    x = rnd((M, Kd), 1.5, torch.float32)
    b = rnd((N, Kd), 0.05)
    c0 = rnd((M, N), dtype=torch.float32)
    c, ss = c0.clone(), torch.zeros(M, device=dev)
    K.gemm_skinny_packed_normed(x, K.pack_weight_frag(b), c, ss, N)
    close(ss, x.pow(2).sum(1), 1e-3 * Kd, 1e-5, f"skinny normed row sums {M}x{Kd}")
    close(c, c0 + x.to(BF).float() @ b.float().t(), 5e-3, 3e-3, f"skinny normed {M}x{N}x{Kd}")


def fuzz_norm(r):
    rows, cols = r.randint(1, 700), r.choice([256, 1280, 1536, 2048, 3584, 5120])
    f32 = bool(r.randint(0, 1))
    x = rnd((rows, cols), 2.0, torch.float32 if f32 else BF)
    w = rnd((cols,)) + 1
    rstd = torch.empty(rows, device=dev)
    y = K.rmsnorm_fwd(x, w, 1e-6, rstd=rstd)
    xr = x.float().clone().requires_grad_(True); wr = w.float().clone().requires_grad_(True)
    want = xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6) * wr
    close(y, want, 3e-2, 2e-2, f"rmsnorm fwd {rows}x{cols}")
    dy = rnd((rows, cols))
    want.backward(dy.float())
    dx = torch.zeros_like(x); dw = torch.zeros(cols, device=dev)
    K.rmsnorm_bwd(x, w, dy, rstd, dx, dw)
    close(dx, xr.grad, 3e-2, 3e-2, f"rmsnorm dx {rows}x{cols} f32={f32}")
    close(dw, wr.grad, 3e-2 * math.sqrt(rows), 2e-2, f"rmsnorm dw {rows}x{cols} f32={f32}")


def fuzz_pair(r):
    """Precise scoring mode (csrc/precise.hip): the pair attention on random segment layouts and the two-pass GEMM with ragged shapes,
    against fp64 torch at ~2^-16 of the output scale (a dropped lo part would miss by two orders of magnitude)."""
    if r.randint(0, 1):
        D, Hq, Hkv = r.choice([(128, 4, 2), (128, 7, 1), (80, 3, 3)])
        causal = D == 128
        segs, at = [], 0
        if causal and r.randint(0, 1):
            P = r.randint(1, 300)
            segs.append((0, P, 0, 0)); at = P
            for _ in range(r.randint(1, 4)):
                L = r.randint(1, 200)
                segs.append((at, L, 0, P)); at += L
        else:
            for _ in range(r.randint(1, 4)):
                L = r.randint(1, 330)
                segs.append((at, L, 0, 0)); at += L
        T = at
        qkv = rnd((T, (Hq + 2 * Hkv) * D), 0.9, torch.float32)
        hi, lo = K.split_pair(qkv)
        qd, kd = Hq * D, Hkv * D
        cut = lambda t: (t[:, :qd], t[:, qd:qd + kd], t[:, qd + kd:])            # noqa: E731
        (qh, kh, vh), (ql, kl, vl) = cut(hi), cut(lo)
        o = K.attn_fwd_pair((qh, ql), (kh, kl), (vh, vl), K.make_segments(segs, dev), max(s[1] for s in segs), Hq, Hkv, D, causal, D ** -0.5)
        x = hi.double() + lo.double()
        q64, k64, v64 = cut(x)
        mask = dense_mask(segs, T, causal).to(dev)
        qq = q64.reshape(T, Hq, D).transpose(0, 1)
        kk = k64.reshape(T, Hkv, D).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
        vv = v64.reshape(T, Hkv, D).repeat_interleave(Hq // Hkv, 1).transpose(0, 1)
        sc = (qq @ kk.transpose(1, 2)) * D ** -0.5
        want = (torch.softmax(sc.masked_fill(~mask, float("-inf")), -1) @ vv).transpose(0, 1).reshape(T, Hq * D)
        got = o[0].double() + o[1].double()
        err = float((got - want).abs().max() / want.abs().max())
        if not err <= 3e-4:
            raise AssertionError(f"MISMATCH attn pair {segs}: rel err {err:.3g}")
    else:
        M, N, Kd = r.randint(1, 3000), r.randint(1, 300) * 8, r.randint(1, 40) * 64
        a = rnd((M, Kd), 1.0, torch.float32)
        w = rnd((N, Kd), 0.05)
        res = rnd((M, N), 1.0, torch.float32)
        want = a.double() @ w.double().t() + res.double()
        x = res.clone()
        K.gemm_pair(*K.split_pair(a), w, residual=x, out=x)
        err = float((x.double() - want).abs().max() / want.abs().max())
        if not err <= 2.0 ** -15:
            raise AssertionError(f"MISMATCH gemm pair {M}x{N}x{Kd}: rel err {err:.3g}")


def fuzz_pair_epilogue(r):
    """Round 5: the pair GEMM with its producer in the epilogue (SwiGLU / rotary / activation + hi/lo split, gemm_halftile.h) on ragged
    M, with and without the K-split tail, against the two-kernel sequence it replaces (pair GEMM -> fp32 -> producer kernel): tapes
    bit-identical, pairs within half the pair precision; and the pair attention's two kernels against each other bit for bit."""
    kind = r.randint(0, 3)
    if kind == 3:
        D, Hq, Hkv = r.choice([(128, 4, 2), (128, 7, 1), (80, 3, 3)])
        causal = D == 128
        segs, at = [], 0
        P = r.randint(1, 400) if causal else 0
        if P:
            segs.append((0, P, 0, 0)); at = P
        for _ in range(r.randint(1, 4)):
            L = r.randint(1, 600)
            segs.append((at, L, 0, P) if P else (at, L, 0, 0)); at += L
        qkv = rnd((at, (Hq + 2 * Hkv) * D), 0.9, torch.float32)
        hi, lo = K.split_pair(qkv)
        qd, kd = Hq * D, Hkv * D
        cut = lambda t: (t[:, :qd], t[:, qd:qd + kd], t[:, qd + kd:])            # noqa: E731
        (qh, kh, vh), (ql, kl, vl) = cut(hi), cut(lo)
        sg = K.make_segments(segs, dev)
        outs = [K.attn_fwd_pair((qh, ql), (kh, kl), (vh, vl), sg, max(x[1] for x in segs), Hq, Hkv, D, causal, D ** -0.5, variant=v) for v in (0, 1)]
        if not (torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])):
            raise AssertionError(f"MISMATCH attn pair variants differ on {segs}")
        return
    M = r.choice([r.randint(1, 600), r.randint(600, 5000)])
    Kd = r.randint(1, 24) * 64
    a = rnd((M, Kd), 1.0, torch.float32)
    ah, al = K.split_pair(a)
    no_split = r.randint(0, 1)
    with K.plan(gemm_tile=256, gemm_no_split=no_split):
        if kind == 0:
            I = r.randint(1, 16) * 128
            w, b = rnd((2 * I, Kd), 0.05), (rnd((2 * I,), 0.3) if r.randint(0, 1) else None)
            tf, tu = torch.empty(M, 2 * I, device=dev, dtype=BF), torch.empty(M, 2 * I, device=dev, dtype=BF)
            f = K.gemm_pair_swiglu(ah, al, w, bias=b, gu_out=tf)
            u = K.swiglu_pair(K.gemm_pair(ah, al, w, bias=b), gu_out=tu)
        elif kind == 1:
            Hq, Hkv = r.randint(1, 12), r.randint(1, 3)
            heads = Hq + 2 * Hkv
            w, b = rnd((heads * 128, Kd), 0.05), rnd((heads * 128,), 0.3)
            ang = torch.rand(M, 64, device=dev) * 6.28
            cos, sin = torch.cat([ang.cos(), ang.cos()], -1).contiguous(), torch.cat([ang.sin(), ang.sin()], -1).contiguous()
            f = K.gemm_pair_rope(ah, al, w, cos, sin, Hq + Hkv, heads, 128, bias=b)
            u = K.rope_pair(K.gemm_pair(ah, al, w, bias=b), cos, sin, Hq + Hkv, heads, 128)
            tf = tu = None
        else:
            N = r.randint(1, 80) * 64
            act = r.choice([K.SPACER_ACT_QUICK_GELU, K.SPACER_ACT_GELU_ERF, K.SPACER_ACT_SILU])
            w, b = rnd((N, Kd), 0.05), rnd((N,), 0.3)
            tf, tu = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev, dtype=BF)
            f = K.gemm_pair_act(ah, al, w, act, bias=b, pre_out=tf)
            u = K.act_pair(K.gemm_pair(ah, al, w, bias=b), act, pre_out=tu)
    if tf is not None:
        # one fp32 summation order (tail off): the same sums, rounded once -> identical tapes.  With the K-split tail on, the SwiGLU
        # form's tiles cover other output columns than the plain form's, so an element may sit in a split tile in one launch and in a
        # whole tile in the other: fp32 association differs, a bf16 rounding may flip (rarely)
        if no_split and not torch.equal(tf, tu):
            raise AssertionError(f"MISMATCH pair epilogue kind {kind} {M}x{Kd}: tape differs with the K-split tail off")
        d = (tf.float() - tu.float()).abs()
        if bool((d > 2.0 ** -7 * tu.float().abs() + 1e-6).any()) or float((tf == tu).float().mean()) < 0.999:
            raise AssertionError(f"MISMATCH pair epilogue kind {kind} {M}x{Kd}: tape off by more than a bf16 ulp ({float(d.max()):.3g})")
    # the two kernels evaluate the same fp32 expression; where the compiler contracted a * b + c differently, a last-bit difference can
    # move a value across a rounding boundary of hi and the pair re-encodes it within its own precision (|lo| <= 2^-8 |x|, lo rounded
    # to 2^-8 of itself: 2^-16 |x| per encoding).  So: element-wise within 2 x 2^-16 |x| (+ a floor for values near zero), and
    # bit-identical almost everywhere -- a wrong partner / bias / table index would miss both by orders of magnitude
    fd, ud = f[0].double() + f[1].double(), u[0].double() + u[1].double()
    scale = float(ud.abs().max())
    bad = (fd - ud).abs() > 2.0 ** -15 * ud.abs() + 2.0 ** -20 * scale
    same = float(((f[0] == u[0]) & (f[1] == u[1])).float().mean())
    if bool(bad.any()) or same < 0.98:
        raise AssertionError(f"MISMATCH pair epilogue kind {kind} {M}x{Kd}: {int(bad.sum())} elements off, max |diff| {float((fd - ud).abs().max()):.3g} at "
                             f"scale {scale:.3g}, {100 * same:.2f} % of the pairs bit-identical")


def fuzz_small_rows(r):
    """Round 5: decode gate|up + SwiGLU for <= 16 rows -- the SMALL instantiation equals the 64-row launch on the same rows bit for bit,
    the norm-folded form equals its own algebra rstd (bf16(x) (W diag w)^T) to bf16 rounding."""
    M, I, Kd = r.randint(1, 16), r.randint(1, 80) * 32, r.randint(1, 16) * 256
    x = rnd((M, Kd), 1.3, torch.float32)
    w = rnd((2 * I, Kd), 0.05)
    lnw = (1.0 + 0.3 * torch.randn(Kd, device=dev)).to(BF)
    h = K.rmsnorm_fwd(x, lnw, 1e-6)
    wp = K.pack_weight_frag_swiglu(w)
    y16 = K.gemm_skinny_swiglu(h, wp, I)
    pad = torch.zeros(17, Kd, device=dev, dtype=BF)
    pad[:M] = h
    if not torch.equal(y16, K.gemm_skinny_swiglu(pad, wp, I)[:M]):
        raise AssertionError(f"MISMATCH small-row swiglu {M}x{I}x{Kd} vs the 64-row launch")
    wnf = (w.float() * lnw.float()[None, :]).to(BF)
    y = K.gemm_skinny_swiglu_normed(x, K.pack_weight_frag_swiglu(wnf), I, 1e-6)
    rstd = torch.rsqrt((x ** 2).mean(1, keepdim=True) + 1e-6)
    gu = (x.to(BF).float() @ wnf.float().t()) * rstd
    close(y, torch.nn.functional.silu(gu[:, :I]) * gu[:, I:], 2e-3, 1e-2, f"norm-folded swiglu {M}x{I}x{Kd}")


def fuzz_chunk_head(r):
    """Vocabulary-chunked log-sum-exp / dlogits (loss.hip) against the one-shot kernels for ragged chunk sizes and strided views."""
    rows, V = r.randint(1, 300), r.randint(3, 500) * 4
    ch = r.choice([V, r.randint(1, max(1, V // 4)) * 4])
    logits = rnd((rows, V + 8), 3.0, torch.float32)[:, 4:V + 4] if r.randint(0, 1) else rnd((rows, V), 3.0, torch.float32)
    tg = torch.randint(0, V, (rows,), device=dev)
    g = rnd((rows,), 1.0, torch.float32)
    state = torch.empty(3, rows, device=dev)
    for c0 in range(0, V, ch):
        K.lse_chunk_(logits[:, c0:min(V, c0 + ch)], tg, c0, state, first=c0 == 0)
    logp, lse = K.lse_finish(state)
    want = torch.log_softmax(logits.double(), -1).gather(1, tg.view(-1, 1)).view(-1)
    close(logp, want, 2e-5, 1e-5, f"chunk lse {rows}x{V} ch {ch}")
    dl = torch.empty(rows, V, device=dev, dtype=BF)
    for c0 in range(0, V, ch):
        c1 = min(V, c0 + ch)
        K.logprob_bwd_chunk(logits[:, c0:c1], tg, c0, lse, g, dl[:, c0:c1])
    p = torch.softmax(logits.double(), -1)
    onehot = torch.zeros_like(p).scatter_(1, tg.view(-1, 1), 1.0)
    close(dl, (onehot - p) * g.double().view(-1, 1), 1e-2 * float(g.abs().max()), 1e-2, f"chunk dlogits {rows}x{V} ch {ch}")


def run(budget: float, seed: int) -> dict:
    r = random.Random(seed)
    torch.manual_seed(seed)
    fns = [fuzz_gemm, fuzz_gemm_trans, fuzz_swiglu, fuzz_attention, fuzz_decode_attention, fuzz_skinny, fuzz_skinny_normed, fuzz_norm, fuzz_resize, fuzz_pair,
           fuzz_chunk_head, fuzz_pair_epilogue, fuzz_small_rows]
    counts = {f.__name__: 0 for f in fns}
    t0 = time.time()
    while time.time() - t0 < budget:
        f = r.choice(fns)
        f(r)
        counts[f.__name__] += 1
    torch.cuda.synchronize()
    return counts


if __name__ == "__main__":
    print("fuzz ok:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 120.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0))
