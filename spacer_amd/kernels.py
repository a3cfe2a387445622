"""Thin torch-tensor front end over the C-ABI (include/spacer_hip.h).

PyTorch is used here for device memory and the current HIP stream only; every function launches
hand-written gfx950 kernels from libspacer_hip.so on ``torch.cuda.current_stream()``.  There is no
eager / CPU fallback: tensors must live on a GPU and the library must be built.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _lib
import contextlib

from ._lib import (SPACER_ACT_GELU_ERF, SPACER_ACT_NONE, SPACER_ACT_QUICK_GELU, SPACER_ACT_SILU,  # noqa: F401
                   AttnSegment, GemmEpilogue, Plan, SpacerError, check)

BF16 = torch.bfloat16


def _plan_from_env() -> Plan:
    """The SPACER_* launch-plan switches (README.md), read ONCE here: libspacer_hip.so reads no environment variables -- every
    launch that has a plan gets this struct (include/spacer_hip.h: spacer_plan).  Tests change it with ``plan(...)``."""
    e = os.environ
    return Plan(gemm_tile=int(e.get("SPACER_GEMM_TILE", "0") or 0), gemm_no_split=int(bool(e.get("SPACER_GEMM_NOSPLIT"))),
                skinny_blocks=int(e.get("SPACER_SKINNY_BLOCKS", "0") or 0), skinny_no_balance=int(bool(e.get("SPACER_SKINNY_NOBALANCE"))),
                cus=int(e.get("SPACER_CUS", "0") or 0), skinny_skew=int(e.get("SPACER_SKINNY_SKEW", "0") or 0))


PLAN = _plan_from_env()
SWIGLU_UNFUSED = bool(os.environ.get("SPACER_GEMM_SWIGLU_UNFUSED"))      # A/B runs: gate|up GEMM + separate SwiGLU launch


def _plan():
    return C.byref(PLAN)


@contextlib.contextmanager
def plan(**fields):
    """Temporarily change launch-plan switches, e.g. ``with K.plan(gemm_no_split=1): ...`` (bit-exact comparisons in tests)."""
    old = {k: getattr(PLAN, k) for k in fields}
    for k, v in fields.items():
        setattr(PLAN, k, int(v))
    try:
        yield PLAN
    finally:
        for k, v in old.items():
            setattr(PLAN, k, v)


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> C.c_void_p:
    if t is None:
        return C.c_void_p(0)
    if not t.is_cuda:
        raise SpacerError("spacer_amd kernels need GPU tensors (there is no CPU path)")
    return C.c_void_p(t.data_ptr())


def _rowmajor(t: torch.Tensor) -> int:
    assert t.dim() == 2 and t.stride(1) == 1, "expected a row-major 2-D view"
    return t.stride(0)


class _Profiler:
    """Live per-kernel timing with HIP events recorded on the launch stream (bench.py's roofline leg).
    Disabled by default; when enabled every profiled launch is bracketed by two events."""

    def __init__(self):
        self.enabled = False
        self.by_shape = False
        self.records = []

    def reset(self, enabled: bool = False):
        self.records = []
        self.enabled = enabled

    def begin(self):
        if not self.enabled:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        return e

    def end(self, name: str, start, flops: float = 0.0, nbytes: float = 0.0):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        self.records.append((name, start, e, flops, nbytes))

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, s, e, flops, nbytes in self.records:
            d = out.setdefault(name, dict(seconds=0.0, flops=0.0, bytes=0.0, launches=0))
            d["seconds"] += s.elapsed_time(e) * 1e-3
            d["flops"] += flops
            d["bytes"] += nbytes
            d["launches"] += 1
        for d in out.values():
            d["tflops"] = d["flops"] / d["seconds"] / 1e12 if d["seconds"] > 0 else 0.0
            d["gbps"] = d["bytes"] / d["seconds"] / 1e9 if d["seconds"] > 0 else 0.0
        return out


PROFILER = _Profiler()


# ----------------------------------------------------------------------------------------- GEMM
_GEMM_WS = {}


def _gemm_workspace(device) -> torch.Tensor:
    """Split-K workspace of the GEMM (fp32 partial-tile slabs), one per device; all launches go through the current
    stream, which is what the C-ABI asks of launches that share a workspace."""
    ws = _GEMM_WS.get(device)
    if ws is None:
        ws = torch.empty(_lib.load().spacer_gemm_workspace_bytes() // 4, device=device, dtype=torch.float32)
        _GEMM_WS[device] = ws
    return ws


def gemm_nt(a: torch.Tensor, b: torch.Tensor, *, out: Optional[torch.Tensor] = None, bias=None, residual=None,
            act: int = SPACER_ACT_NONE, out_dtype=BF16, alpha: float = 1.0, algo_k: Optional[int] = None,
            split_k: bool = True) -> torch.Tensor:
    """out[M,N] = act(alpha * a[M,K] @ b[N,K]^T + bias) + residual.  a, b bf16; out bf16 or fp32."""
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and a.dtype == BF16 and b.dtype == BF16
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    assert out.dtype in (BF16, torch.float32)
    if residual is not None:
        assert residual.dtype == out.dtype
    epi = GemmEpilogue(_ptr(bias), _ptr(residual), _rowmajor(residual) if residual is not None else 0,
                       1 if out.dtype == torch.float32 else 0, act, alpha)
    if split_k:
        ws = _gemm_workspace(a.device)
        epi.workspace, epi.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    epi.plan = C.pointer(PLAN)
    t0 = PROFILER.begin()
    check(_lib.load().spacer_gemm_bf16_nt(_ptr(a), _rowmajor(a), _ptr(b), _rowmajor(b), _ptr(out), _rowmajor(out),
                                          M, N, K, C.byref(epi), _stream()), "gemm_bf16_nt")
    ka = algo_k or K       # algorithmic contraction length (dW GEMMs run on a zero-padded token dim)
    if t0 is not None:
        # rocprof names the 256-tile instantiations <BALANCED, TA, TB, STG16>; STG16 = bf16 output without a residual (bf16 staging
        # epilogue, persistent workgroups)
        stg = "true" if (out.dtype == BF16 and residual is None) else "false"
        name = (f"gemm_bf16_nt_256h_kernel<true, false, false, {stg}>"
                if _lib.load().spacer_gemm_tile(M, N, K, 1 if split_k else 0, _plan()) == 256 else "gemm_bf16_nt_kernel")
        PROFILER.end(name, t0, 2.0 * M * N * ka, 2.0 * (M * ka + N * ka) + out.element_size() * M * N)
    if PROFILER.by_shape and t0 is not None:
        s0, e0 = PROFILER.records[-1][1], PROFILER.records[-1][2]
        PROFILER.records.append((f"gemm[{M}x{N}x{K}]", s0, e0, 2.0 * M * N * ka, 0.0))
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, *, trans_a: bool = False, trans_b: bool = False, out: Optional[torch.Tensor] = None,
         residual=None, out_dtype=BF16, alpha: float = 1.0) -> torch.Tensor:
    """out[M,N] = alpha * sum_k A(m,k) B(n,k) + residual with operands read IN PLACE from contraction-major arrays:
    trans_a -> ``a`` is [K, M] (A(m,k) = a[k, m]), trans_b -> ``b`` is [K, N].  The backward GEMMs of every linear layer:
        dX = gemm(dY, W, trans_b=True)                    [T, out] x [out, in] -> [T, in]
        dW = gemm(dY, X, trans_a=True, trans_b=True)      [T, out], [T, in]    -> [out, in]   (K = T, ragged is fine)
    No transposes, no W^T copies.  Runs on the 256 tile (spacer_gemm_bf16)."""
    if not trans_a and not trans_b:
        return gemm_nt(a, b, out=out, residual=residual, out_dtype=out_dtype, alpha=alpha)
    Kc = a.shape[0] if trans_a else a.shape[1]
    M = a.shape[1] if trans_a else a.shape[0]
    N = b.shape[1] if trans_b else b.shape[0]
    assert (b.shape[0] if trans_b else b.shape[1]) == Kc and a.dtype == BF16 and b.dtype == BF16
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    assert out.dtype in (BF16, torch.float32) and (residual is None or residual.dtype == out.dtype)
    epi = GemmEpilogue(None, _ptr(residual), _rowmajor(residual) if residual is not None else 0,
                       1 if out.dtype == torch.float32 else 0, SPACER_ACT_NONE, alpha)
    ws = _gemm_workspace(a.device)
    epi.workspace, epi.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    epi.plan = C.pointer(PLAN)
    t0 = PROFILER.begin()
    check(_lib.load().spacer_gemm_bf16(_ptr(a), _rowmajor(a), _ptr(b), _rowmajor(b), _ptr(out), _rowmajor(out), M, N, Kc,
                                       int(trans_a), int(trans_b), C.byref(epi), _stream()), "gemm_bf16")
    if t0 is not None:
        stg = "true" if (out.dtype == BF16 and residual is None) else "false"
        name = f"gemm_bf16_nt_256h_kernel<true, {'true' if trans_a else 'false'}, true, {stg}>"
        rmw = out.element_size() * M * N if residual is not None else 0
        PROFILER.end(name, t0, 2.0 * M * N * Kc, 2.0 * (M * Kc + N * Kc) + out.element_size() * M * N + rmw)
        if PROFILER.by_shape:
            s0, e0 = PROFILER.records[-1][1], PROFILER.records[-1][2]
            PROFILER.records.append((f"gemm[{'dW' if trans_a else 'dX'} {M}x{N}x{Kc}]", s0, e0, 2.0 * M * N * Kc, 0.0))
    return out


def gemm_swiglu(a: torch.Tensor, w_gu: torch.Tensor, *, bias=None, keep_gu: bool = True, out=None, gu_out=None):
    """SwiGLU input half of an MLP: returns (act [M, I] bf16, gu [M, 2I] bf16 or None) with
    act = silu(a @ Wgate^T + b) * (a @ Wup^T + b), w_gu = [gate rows | up rows] ([2I, K]).  One launch (gate|up GEMM with the
    SwiGLU in its epilogue) when the shape runs on the 256 tile, else GEMM + swiglu_fwd; the bits are the same either way.
    ``keep_gu=False`` (no backward: reference model, rollout prefill) skips writing gate|up in the fused form."""
    M, K = a.shape
    two_i = w_gu.shape[0]
    inter = two_i // 2
    if not _lib.load().spacer_gemm_swiglu_fused(M, inter, K, _plan()) or SWIGLU_UNFUSED:
        gu = gemm_nt(a, w_gu, bias=bias, out=gu_out if keep_gu else None)
        return swiglu_fwd(gu, out=out), (gu if keep_gu else None)
    act = out if out is not None else torch.empty(M, inter, device=a.device, dtype=BF16)
    gu = (gu_out if gu_out is not None else torch.empty(M, two_i, device=a.device, dtype=BF16)) if keep_gu else None
    t0 = PROFILER.begin()
    check(_lib.load().spacer_gemm_swiglu_bf16(_ptr(a), _rowmajor(a), _ptr(w_gu), _rowmajor(w_gu), _ptr(bias), _ptr(act), _rowmajor(act),
                                              _ptr(gu), _rowmajor(gu) if gu is not None else 0, M, inter, K, _stream()),
          "gemm_swiglu_bf16")
    if t0 is not None:
        PROFILER.end("gemm_bf16_nt_256h_kernel<true, false, false, true>", t0, 2.0 * M * two_i * K,
                     2.0 * (M * K + two_i * K) + 2.0 * M * (inter + (two_i if keep_gu else 0)))
        if PROFILER.by_shape:
            s0, e0 = PROFILER.records[-1][1], PROFILER.records[-1][2]
            PROFILER.records.append((f"gemm_swiglu[{M}x{two_i}x{K}]", s0, e0, 2.0 * M * two_i * K, 0.0))
    return act, gu


def gemm_skinny_acc(a: torch.Tensor, b: torch.Tensor, c32: torch.Tensor) -> torch.Tensor:
    """c32[M,N] (fp32) += a[M,K] @ b[N,K]^T for M <= 64 (decode)."""
    M, K = a.shape
    N = b.shape[0]
    assert c32.dtype == torch.float32 and a.dtype == BF16 and b.dtype == BF16
    epi = GemmEpilogue(None, _ptr(c32), _rowmajor(c32), 1, SPACER_ACT_NONE, 1.0)
    epi.plan = C.pointer(PLAN)
    check(_lib.load().spacer_gemm_skinny_bf16(_ptr(a), _rowmajor(a), _ptr(b), _rowmajor(b), _ptr(c32), _rowmajor(c32),
                                              M, N, K, C.byref(epi), _stream()), "gemm_skinny_bf16")
    return c32


def pack_weight_frag(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[N, K] row-major bf16 -> fragment-major flat copy for gemm_skinny_packed_acc."""
    N, K = w.shape
    if out is None:
        out = torch.empty(N * K, device=w.device, dtype=BF16)
    check(_lib.load().spacer_pack_weight_frag(_ptr(w), _rowmajor(w), _ptr(out), N, K, _stream()), "pack_weight_frag")
    return out


def gemm_skinny_packed_acc(a: torch.Tensor, bp: torch.Tensor, c32: torch.Tensor, N: int) -> torch.Tensor:
    """c32[M,N] (fp32) += a[M,K] @ W[N,K]^T with W given in packed (fragment-major) form."""
    M, K = a.shape
    assert bp.numel() == N * K and c32.dtype == torch.float32
    check(_lib.load().spacer_gemm_skinny_packed_bf16(_ptr(a), _rowmajor(a), _ptr(bp), _ptr(c32), _rowmajor(c32), M, N, K,
                                                     _plan(), _stream()), "gemm_skinny_packed_bf16")
    return c32


def gemm_skinny_packed_normed(x32: torch.Tensor, bp: torch.Tensor, c32: torch.Tensor, rowss: torch.Tensor, N: int) -> torch.Tensor:
    """c32[M,N] (fp32) += bf16(x32[M,K]) @ Wp^T and rowss[m] += sum_k x32[m,k]^2: the decode projection with the RMSNorm in front
    of it folded in (Wp = pack_weight_frag(W * w_norm)); decode_qkv_finish_normed applies rstd."""
    M, K = x32.shape
    assert bp.numel() == N * K and c32.dtype == torch.float32 and x32.dtype == torch.float32 and rowss.dtype == torch.float32
    check(_lib.load().spacer_gemm_skinny_packed_normed(_ptr(x32), _rowmajor(x32), _ptr(bp), _ptr(c32), _rowmajor(c32), _ptr(rowss),
                                                       M, N, K, _plan(), _stream()), "gemm_skinny_packed_normed")
    return c32


def gemm_skinny_packed_store(a: torch.Tensor, bp: torch.Tensor, c32: torch.Tensor, N: int) -> torch.Tensor:
    """c32[M,N] (fp32) = a[M,K] @ W[N,K]^T (store, no accumulate) for wide N (lm_head): no zero fill of c32 needed."""
    M, K = a.shape
    assert bp.numel() == N * K and c32.dtype == torch.float32
    check(_lib.load().spacer_gemm_skinny_packed_store_bf16(_ptr(a), _rowmajor(a), _ptr(bp), _ptr(c32), _rowmajor(c32), M, N, K,
                                                           _plan(), _stream()), "gemm_skinny_packed_store_bf16")
    return c32


def pack_weight_frag_swiglu(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[2*I, K] = [gate | up] rows -> fragment-major copy with 8 gate + 8 up columns per fragment (gemm_skinny_swiglu)."""
    N, K = w.shape
    if out is None:
        out = torch.empty(N * K, device=w.device, dtype=BF16)
    check(_lib.load().spacer_pack_weight_frag_swiglu(_ptr(w), _rowmajor(w), _ptr(out), N // 2, K, _stream()), "pack_weight_frag_swiglu")
    return out


_SWIGLU_WS = {}


def gemm_skinny_swiglu(a: torch.Tensor, bp: torch.Tensor, inter: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, inter] (bf16) = silu(a @ Wgate^T) * (a @ Wup^T), weights from pack_weight_frag_swiglu.  The per-device
    workspace (zeroed once, left zeroed by the kernel) lets the launch balance its last round of column groups."""
    M, K = a.shape
    assert bp.numel() == 2 * inter * K
    if out is None:
        out = torch.empty(M, inter, device=a.device, dtype=BF16)
    ws = _SWIGLU_WS.get(a.device)
    if ws is None:
        ws = torch.zeros(_lib.load().spacer_gemm_skinny_swiglu_workspace_bytes() // 4, device=a.device, dtype=torch.int32)
        _SWIGLU_WS[a.device] = ws
    check(_lib.load().spacer_gemm_skinny_swiglu_bf16_ws(_ptr(a), _rowmajor(a), _ptr(bp), _ptr(out), _rowmajor(out), M, inter, K,
                                                        _ptr(ws), ws.numel() * 4, _plan(), _stream()), "gemm_skinny_swiglu_bf16")
    return out


def gemm_skinny_swiglu_normed(x32: torch.Tensor, bp: torch.Tensor, inter: int, eps: float, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[M, inter] = silu(rstd g) * (rstd u) with [g | u] = bf16(x32) @ Wp^T, rstd = rsqrt(mean(x32^2) + eps): the decode gate|up
    projection of <= 16 rows with the RMSNorm in front of it folded in; bp = pack_weight_frag_swiglu(W * w_norm[None, :])."""
    M, K = x32.shape
    assert x32.dtype == torch.float32 and M <= 16 and bp.numel() == 2 * inter * K
    if out is None:
        out = torch.empty(M, inter, device=x32.device, dtype=BF16)
    ws = _SWIGLU_WS.get(x32.device)
    if ws is None:
        ws = torch.zeros(_lib.load().spacer_gemm_skinny_swiglu_workspace_bytes() // 4, device=x32.device, dtype=torch.int32)
        _SWIGLU_WS[x32.device] = ws
    check(_lib.load().spacer_gemm_skinny_swiglu_normed(_ptr(x32), _rowmajor(x32), _ptr(bp), _ptr(out), _rowmajor(out), M, inter, K, eps,
                                                       _ptr(ws), ws.numel() * 4, _plan(), _stream()), "gemm_skinny_swiglu_normed")
    return out


def transpose_pad(x: torch.Tensor, rpad: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x[R,C] bf16 -> out[C, Rpad] with zero fill (Rpad defaults to R rounded up to 64)."""
    R, Cc = x.shape
    if rpad is None:
        rpad = (R + 63) // 64 * 64
    if out is None:
        out = torch.empty(Cc, rpad, device=x.device, dtype=BF16)
    check(_lib.load().spacer_transpose_bf16(_ptr(x), _rowmajor(x), _ptr(out), _rowmajor(out), R, Cc, rpad, _stream()),
          "transpose_bf16")
    return out


# ----------------------------------------------------------------------------------------- norms
def rmsnorm_fwd(x, w, eps, *, rstd=None, out=None):
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=BF16)
    check(_lib.load().spacer_rmsnorm_fwd(_ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(out), _ptr(rstd), rows,
                                         cols, eps, _stream()), "rmsnorm_fwd")
    return out


def rmsnorm_bwd(x, w, dy, rstd, dx, dw, *, accumulate=True):
    rows, cols = x.shape
    assert dx.dtype == x.dtype and dw.dtype == torch.float32
    ws = _gemm_workspace(x.device)          # two-stage dw reduction (no atomics); launches on one stream share the scratch
    check(_lib.load().spacer_rmsnorm_bwd_ws(_ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(dy), _ptr(rstd), _ptr(dx),
                                            int(accumulate), _ptr(dw), rows, cols, _ptr(ws), ws.numel() * 4, _stream()), "rmsnorm_bwd")
    return dx


def layernorm_fwd(x, w, b, eps=1e-6, *, mean=None, rstd=None, out=None):
    rows, cols = x.shape
    if out is None:
        out = torch.empty(rows, cols, device=x.device, dtype=BF16)
    check(_lib.load().spacer_layernorm_fwd(_ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(b), _ptr(out), _ptr(mean),
                                           _ptr(rstd), rows, cols, eps, _stream()), "layernorm_fwd")
    return out


def layernorm_bwd(x, w, dy, mean, rstd, dx, dw, db, *, accumulate=True):
    rows, cols = x.shape
    ws = _gemm_workspace(x.device)
    check(_lib.load().spacer_layernorm_bwd_ws(_ptr(x), int(x.dtype == torch.float32), _ptr(w), _ptr(dy), _ptr(mean), _ptr(rstd),
                                              _ptr(dx), int(accumulate), _ptr(dw), _ptr(db), rows, cols, _ptr(ws), ws.numel() * 4,
                                              _stream()), "layernorm_bwd")
    return dx


# ----------------------------------------------------------------------------------------- rotary
def rope_(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, head_dim: int, *, inverse=False):
    """In-place rotary on the first ``heads`` heads of every token row of x (2-D [tokens, >= heads*head_dim])."""
    tokens = x.shape[0]
    assert cos.dtype == torch.float32 and cos.shape == (tokens, head_dim) and cos.is_contiguous() and sin.is_contiguous()
    check(_lib.load().spacer_rope_inplace(_ptr(x), x.stride(0), _ptr(cos), _ptr(sin), tokens, heads, head_dim,
                                          int(inverse), _stream()), "rope_inplace")
    return x


# ----------------------------------------------------------------------------------------- attention
def make_segments(segs: Sequence[Sequence[int]], device) -> torch.Tensor:
    """[(q_start, q_len, pre_start, pre_len), ...] -> int32 device tensor [n, 4]."""
    return torch.tensor(list(segs), dtype=torch.int32, device=device).reshape(-1, 4).contiguous()


def attn_fwd(q, k, v, segs: torch.Tensor, max_q_len: int, Hq: int, Hkv: int, D: int, causal: bool, scale: float,
             *, out=None, lse=None):
    """q [T, >=Hq*D] view, k/v [T, >=Hkv*D] views (row stride = token stride).  Returns (o [T,Hq*D], lse [Hq,T])."""
    T = q.shape[0]
    if out is None:
        out = torch.empty(T, Hq * D, device=q.device, dtype=BF16)
    if lse is None:
        lse = torch.empty(Hq, T, device=q.device, dtype=torch.float32)
    assert k.stride(0) == v.stride(0)
    check(_lib.load().spacer_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(out), _ptr(lse), q.stride(0), k.stride(0),
                                      out.stride(0), _ptr(segs), segs.shape[0], max_q_len, T, Hq, Hkv, D, int(causal),
                                      scale, _stream()), "attn_fwd")
    return out, lse


def attn_bwd(q, k, v, o, d_o, lse, segs, max_q_len, Hq, Hkv, D, causal, scale, *, dq, dk32, dv32, delta=None):
    """dq bf16 in q's layout; dk32/dv32 fp32 [T, Hkv*D] zeroed by the caller."""
    T = q.shape[0]
    if delta is None:
        delta = torch.empty(Hq, T, device=q.device, dtype=torch.float32)
    assert o.stride(0) == d_o.stride(0) and dq.stride(0) == q.stride(0) and k.stride(0) == v.stride(0)
    assert dk32.is_contiguous() and dv32.is_contiguous() and dk32.dtype == torch.float32
    check(_lib.load().spacer_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(d_o), _ptr(lse), _ptr(delta), _ptr(dq),
                                      _ptr(dk32), _ptr(dv32), q.stride(0), k.stride(0), o.stride(0), _ptr(segs),
                                      segs.shape[0], max_q_len, T, Hq, Hkv, D, int(causal), scale, _stream()), "attn_bwd")
    return dq, dk32, dv32


def attn_decode(q, prefix_k, prefix_v, prefix_len, prompt_of, tail_k, tail_v, tail_len_dev, Hq, Hkv, D, scale, *, out=None):
    B = q.shape[0]
    if out is None:
        out = torch.empty(B, Hq * D, device=q.device, dtype=BF16)
    check(_lib.load().spacer_attn_decode(_ptr(q), _ptr(prefix_k), _ptr(prefix_v), _ptr(prefix_len), _ptr(prompt_of),
                                         _ptr(tail_k), _ptr(tail_v), _ptr(tail_len_dev), _ptr(out), B, prefix_k.shape[1],
                                         tail_k.shape[1], Hq, Hkv, D, scale, _stream()), "attn_decode")
    return out


def attn_decode_workspace_bytes(n_prompts: int, Hkv: int) -> int:
    return int(_lib.load().spacer_attn_decode_workspace_bytes(n_prompts, Hkv))


def attn_decode_shared(q, prefix_k, prefix_v, prefix_len, prompt_of, tail_k, tail_v, tail_len_dev, Kn, Hq, Hkv, D, scale, *,
                       out=None, workspace=None):
    """attn_decode with the prompt keys scored once per prompt for its Kn rollouts (rows b = prompt*Kn + k)."""
    B = q.shape[0]
    if out is None:
        out = torch.empty(B, Hq * D, device=q.device, dtype=BF16)
    if workspace is None:
        workspace = torch.empty(_lib.load().spacer_attn_decode_workspace_bytes(prefix_k.shape[0], Hkv) // 4, device=q.device,
                                dtype=torch.float32)
    check(_lib.load().spacer_attn_decode_shared(_ptr(q), _ptr(prefix_k), _ptr(prefix_v), _ptr(prefix_len), _ptr(prompt_of),
                                                _ptr(tail_k), _ptr(tail_v), _ptr(tail_len_dev), _ptr(out), _ptr(workspace), B, Kn,
                                                prefix_k.shape[1], tail_k.shape[1], Hq, Hkv, D, scale, _stream()),
          "attn_decode_shared")
    return out


def attn_decode_shared_rows(q, prefix_k, prefix_v, prefix_len, prompt_of, row0, tail_k, tail_v, tail_len_dev, Kmax, Hq, Hkv, D, scale, *,
                            out=None, workspace=None):
    """attn_decode_shared with per-prompt rollout counts: prompt p owns rows [row0[p], row0[p + 1]) (int32 [n_prompts + 1]), <= Kmax each."""
    B, nP = q.shape[0], prefix_k.shape[0]
    assert row0.dtype == torch.int32 and row0.numel() == nP + 1 and prompt_of.numel() == B
    if out is None:
        out = torch.empty(B, Hq * D, device=q.device, dtype=BF16)
    if workspace is None:
        workspace = torch.empty(_lib.load().spacer_attn_decode_workspace_bytes(nP, Hkv) // 4, device=q.device, dtype=torch.float32)
    check(_lib.load().spacer_attn_decode_shared_rows(_ptr(q), _ptr(prefix_k), _ptr(prefix_v), _ptr(prefix_len), _ptr(prompt_of), _ptr(row0),
                                                     _ptr(tail_k), _ptr(tail_v), _ptr(tail_len_dev), _ptr(out), _ptr(workspace), B, nP, Kmax,
                                                     prefix_k.shape[1], tail_k.shape[1], Hq, Hkv, D, scale, _stream()),
          "attn_decode_shared_rows")
    return out


# ----------------------------------------------------------------------------------------- element-wise
def swiglu_fwd(gu, *, out=None):
    rows, two_i = gu.shape
    if out is None:
        out = torch.empty(rows, two_i // 2, device=gu.device, dtype=BF16)
    check(_lib.load().spacer_swiglu_fwd(_ptr(gu), _ptr(out), rows, two_i // 2, _stream()), "swiglu_fwd")
    return out


def swiglu_bwd(gu, dy, *, out=None):
    rows, two_i = gu.shape
    if out is None:
        out = torch.empty_like(gu)
    check(_lib.load().spacer_swiglu_bwd(_ptr(gu), _ptr(dy), _ptr(out), rows, two_i // 2, _stream()), "swiglu_bwd")
    return out


def act_fwd(x, act, *, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().spacer_act_fwd(_ptr(x), _ptr(out), x.numel(), act, _stream()), "act_fwd")
    return out


def act_bwd(x, dy, act, *, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().spacer_act_bwd(_ptr(x), _ptr(dy), _ptr(out), x.numel(), act, _stream()), "act_bwd")
    return out


def bias_grad_(dy, db32):
    rows, cols = dy.shape
    check(_lib.load().spacer_bias_grad(_ptr(dy), _rowmajor(dy), _ptr(db32), rows, cols, _stream()), "bias_grad")
    return db32


def zero_(t: torch.Tensor) -> torch.Tensor:
    """In-place zero fill of a contiguous device tensor on the current stream (spacer_zero)."""
    assert t.is_contiguous()
    check(_lib.load().spacer_zero(_ptr(t), t.numel() * t.element_size(), _stream()), "zero")
    return t


def zeros(*shape, device, dtype=torch.float32) -> torch.Tensor:
    return zero_(torch.empty(*shape, device=device, dtype=dtype))


def cast_bf16(x32, *, out=None):
    if out is None:
        out = torch.empty(x32.shape, device=x32.device, dtype=BF16)
    check(_lib.load().spacer_cast_f32_to_bf16(_ptr(x32), _ptr(out), x32.numel(), _stream()), "cast_f32_to_bf16")
    return out


def cast_bf16_strided(x32, out):
    rows, cols = x32.shape
    check(_lib.load().spacer_cast_f32_to_bf16_strided(_ptr(x32), _rowmajor(x32), _ptr(out), _rowmajor(out), rows, cols,
                                                      _stream()), "cast_f32_to_bf16_strided")
    return out


def cast_f32(x, *, out=None):
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    check(_lib.load().spacer_cast_bf16_to_f32(_ptr(x), _ptr(out), x.numel(), _stream()), "cast_bf16_to_f32")
    return out


def gather_rows(src, idx32, *, out=None):
    n, cols = idx32.shape[0], src.shape[1]
    if out is None:
        out = torch.empty(n, cols, device=src.device, dtype=BF16)
    check(_lib.load().spacer_gather_rows_bf16(_ptr(src), _rowmajor(src), _ptr(idx32), _ptr(out), n, cols, _stream()),
          "gather_rows_bf16")
    return out


def scatter_add_rows_(src, idx32, dst32):
    n, cols = src.shape
    assert src.is_contiguous() and dst32.dtype == torch.float32
    check(_lib.load().spacer_scatter_add_rows_f32(_ptr(src), _ptr(idx32), _ptr(dst32), _rowmajor(dst32), n, cols, _stream()),
          "scatter_add_rows_f32")
    return dst32


def embed_fwd(ids, table, video, video_row_of_token, *, out=None):
    T, H = ids.shape[0], table.shape[1]
    if out is None:
        out = torch.empty(T, H, device=table.device, dtype=torch.float32)
    check(_lib.load().spacer_embed_fwd(_ptr(ids), _ptr(table), _ptr(video), _ptr(video_row_of_token), _ptr(out), T, H,
                                       _stream()), "embed_fwd")
    return out


def embed_bwd(ids, video_row_of_token, d_out, d_table32, d_video32):
    T, H = d_out.shape
    check(_lib.load().spacer_embed_bwd(_ptr(ids), _ptr(video_row_of_token), _ptr(d_out), _ptr(d_table32), _ptr(d_video32),
                                       T, H, _stream()), "embed_bwd")


def patchify(frames_u8, patch=14, tpatch=2, merge=2, kpad=1216, *, out=None):
    F, Cc, Hpx, Wpx = frames_u8.shape
    assert Cc == 3 and frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    gt, gh, gw = (F + tpatch - 1) // tpatch, Hpx // patch, Wpx // patch
    if out is None:
        out = torch.empty(gt * gh * gw, kpad, device=frames_u8.device, dtype=BF16)
    check(_lib.load().spacer_patchify(_ptr(frames_u8), _ptr(out), F, Hpx, Wpx, patch, tpatch, merge, kpad, _stream()),
          "patchify")
    return out, (gt, gh, gw)


_AA_DEV = {}


def resize_bicubic_aa(frames_u8: torch.Tensor, hw, tables_x, tables_y, *, out=None) -> torch.Tensor:
    """uint8 [F, C, H, W] -> uint8 [F, C, h, w], bicubic + antialias with torch's arithmetic (csrc/frontend.hip); the window /
    weight tables (qwen_vl_utils.vision_process.aa_tables) are uploaded once per (size, device)."""
    F, Cc, H, W = frames_u8.shape
    h, w = int(hw[0]), int(hw[1])
    assert frames_u8.dtype == torch.uint8 and frames_u8.is_contiguous()
    dev = frames_u8.device
    key = (dev, H, W, h, w)
    tb = _AA_DEV.get(key)
    if tb is None:
        tb = [t.to(dev) for t in tables_x] + [t.to(dev) for t in tables_y]
        _AA_DEV[key] = tb
    xmin, xsize, wx, ymin, ysize, wy = tb
    if out is None:
        out = torch.empty(F, Cc, h, w, device=dev, dtype=torch.uint8)
    ws = torch.empty(F * Cc * H * w, device=dev, dtype=torch.float32)
    check(_lib.load().spacer_resize_bicubic_aa_u8(_ptr(frames_u8), _ptr(out), F * Cc, H, W, h, w, _ptr(xmin), _ptr(xsize), _ptr(wx),
                                                  wx.shape[1], _ptr(ymin), _ptr(ysize), _ptr(wy), wy.shape[1], _ptr(ws), _stream()),
          "resize_bicubic_aa_u8")
    return out


def gather_frames(frames_u8: torch.Tensor, idx32: torch.Tensor, *, out=None) -> torch.Tensor:
    """out[f] = frames[idx[f]] (uniform frame sampling on the device); frames [T, ...] uint8 contiguous, 16-byte-multiple frames."""
    n = idx32.shape[0]
    fb = frames_u8[0].numel()
    if out is None:
        out = torch.empty((n,) + tuple(frames_u8.shape[1:]), device=frames_u8.device, dtype=torch.uint8)
    check(_lib.load().spacer_gather_frames_u8(_ptr(frames_u8), _ptr(idx32), _ptr(out), n, fb, _stream()), "gather_frames_u8")
    return out


# ----------------------------------------------------------------------------------------- precise scoring mode (csrc/precise.hip)
def _pair_out(rows, cols, device):
    return (torch.empty(rows, cols, device=device, dtype=BF16), torch.empty(rows, cols, device=device, dtype=BF16))


PAIR_TWOPASS = bool(os.environ.get("SPACER_GEMM_PAIR_TWOPASS"))        # A/B runs: two accumulate passes instead of the K-concatenated launch


def gemm_pair(a_hi, a_lo, w, *, bias=None, residual=None, out=None):
    """fp32 out = (a_hi + a_lo) @ w^T + bias + residual (weights are exactly bf16, so only the activation operand is a pair).
    ONE launch over the K-concatenated operands [a_hi | a_lo] . [w | w]^T where the 256-tile kernel takes the shape
    (spacer_gemm_bf16_pair_nt: the fp32 output is written once), else two accumulate passes of the production GEMM.  ``out`` may
    alias ``residual`` (in-place update of the stream)."""
    M, Kd = a_hi.shape
    N = w.shape[0]
    lib = _lib.load()
    if (not PAIR_TWOPASS and a_hi.stride(0) == a_lo.stride(0) and tuple(a_lo.shape) == (M, Kd)
            and lib.spacer_gemm_pair_fused(M, N, Kd, 1, _plan())):
        if out is None:
            out = torch.empty(M, N, device=a_hi.device, dtype=torch.float32)
        assert out.dtype == torch.float32 and (residual is None or residual.dtype == torch.float32)
        epi = GemmEpilogue(_ptr(bias), _ptr(residual), _rowmajor(residual) if residual is not None else 0, 1, SPACER_ACT_NONE, 1.0)
        ws = _gemm_workspace(a_hi.device)
        epi.workspace, epi.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        epi.plan = C.pointer(PLAN)
        t0 = PROFILER.begin()
        check(lib.spacer_gemm_bf16_pair_nt(_ptr(a_hi), _ptr(a_lo), _rowmajor(a_hi), _ptr(w), _rowmajor(w), _ptr(out), _rowmajor(out),
                                           M, N, Kd, C.byref(epi), _stream()), "gemm_bf16_pair_nt")
        if t0 is not None:
            PROFILER.end("gemm_bf16_pair_256h_kernel<1>", t0, 4.0 * M * N * Kd, 2.0 * (2 * M * Kd + N * Kd) + 4.0 * M * N)
        return out
    out = gemm_nt(a_hi, w, bias=bias, residual=residual, out=out, out_dtype=torch.float32)
    return gemm_nt(a_lo, w, residual=out, out=out, out_dtype=torch.float32)


PAIR_EPILOGUE_UNFUSED = bool(os.environ.get("SPACER_PAIR_EPILOGUE_UNFUSED"))   # A/B runs: pair GEMM + separate producer kernel (round 4's path)


def _pair_epilogue(kind, a_hi, a_lo, w, bias, n_out, *, tape=None, cos=None, sin=None, rope_heads=0, head_dim=0, act=SPACER_ACT_NONE):
    """One launch of spacer_gemm_bf16_pair_epilogue (the pair GEMM with its producer in the epilogue); None when the problem does not
    run fused (small shapes on the 128 tile, head_dim != 128, operands with different strides, or the A/B switch)."""
    M, Kd = a_hi.shape
    N = w.shape[0]
    lib = _lib.load()
    if (PAIR_EPILOGUE_UNFUSED or PAIR_TWOPASS or a_hi.stride(0) != a_lo.stride(0) or tuple(a_lo.shape) != (M, Kd)
            or not lib.spacer_gemm_pair_epilogue_fused(kind, M, N, Kd, head_dim, 1, _plan())):
        return None
    hi, lo = _pair_out(M, n_out, a_hi.device)
    ws = _gemm_workspace(a_hi.device)
    t0 = PROFILER.begin()
    check(lib.spacer_gemm_bf16_pair_epilogue(kind, _ptr(a_hi), _ptr(a_lo), _rowmajor(a_hi), _ptr(w), _rowmajor(w), _ptr(bias), _ptr(hi), _ptr(lo),
                                             n_out, _ptr(tape), _rowmajor(tape) if tape is not None else 0, _ptr(cos), _ptr(sin), rope_heads,
                                             head_dim, act, M, N, Kd, ws.data_ptr(), ws.numel() * 4, _plan(), _stream()),
          "gemm_bf16_pair_epilogue")
    if t0 is not None:
        PROFILER.end(f"gemm_bf16_pair_256h_kernel<{kind}>", t0, 4.0 * M * N * Kd, 2.0 * (2 * M * Kd + N * Kd) + 4.0 * M * n_out)
    return hi, lo


def gemm_pair_swiglu(a_hi, a_lo, w_gu, *, bias=None, gu_out=None):
    """(hi, lo) pair of silu(g) * u, [g | u] = (a_hi + a_lo) @ w_gu^T + bias in fp32 -- ONE launch when the 256 tile takes the shape
    (SwiGLU + hi/lo split in the pair GEMM's epilogue; the fp32 [rows, 2I] tensor is never written), else gemm_pair + swiglu_pair.
    ``gu_out`` (bf16 [rows, 2I]) also receives bf16(g | u), the point swiglu_bwd differentiates at."""
    two_i = w_gu.shape[0]
    assert gu_out is None or (gu_out.dtype == BF16 and gu_out.stride(1) == 1 and tuple(gu_out.shape) == (a_hi.shape[0], two_i))
    out = _pair_epilogue(_lib.SPACER_PAIR_SWIGLU, a_hi, a_lo, w_gu, bias, two_i // 2, tape=gu_out) if two_i % 256 == 0 else None
    if out is not None:
        return out
    return swiglu_pair(gemm_pair(a_hi, a_lo, w_gu, bias=bias), gu_out=gu_out)


def gemm_pair_rope(a_hi, a_lo, w, cos, sin, rot_heads, heads, head_dim, *, bias=None):
    """(hi, lo) pair of the rotary-embedded q | k | v rows of (a_hi + a_lo) @ w^T + bias: bias + rotary + split in the pair GEMM's
    epilogue when head_dim == 128 and the 256 tile takes the shape, else gemm_pair + rope_pair."""
    assert w.shape[0] == heads * head_dim
    out = None
    if head_dim == 128:
        assert cos.dtype == torch.float32 and tuple(cos.shape) == (a_hi.shape[0], 128) and cos.is_contiguous() and sin.is_contiguous()
        out = _pair_epilogue(_lib.SPACER_PAIR_ROPE, a_hi, a_lo, w, bias, heads * head_dim, cos=cos, sin=sin, rope_heads=rot_heads, head_dim=128)
    if out is not None:
        return out
    return rope_pair(gemm_pair(a_hi, a_lo, w, bias=bias), cos, sin, rot_heads, heads, head_dim)


def gemm_pair_act(a_hi, a_lo, w, act, *, bias=None, pre_out=None):
    """(hi, lo) pair of act((a_hi + a_lo) @ w^T + bias); ``pre_out`` (bf16) also receives bf16 of the pre-activation (act_bwd's input)."""
    N = w.shape[0]
    assert pre_out is None or (pre_out.dtype == BF16 and pre_out.stride(1) == 1 and tuple(pre_out.shape) == (a_hi.shape[0], N))
    out = _pair_epilogue(_lib.SPACER_PAIR_ACT, a_hi, a_lo, w, bias, N, tape=pre_out, act=act)
    if out is not None:
        return out
    return act_pair(gemm_pair(a_hi, a_lo, w, bias=bias), act, pre_out=pre_out)


def split_pair(x32):
    rows, cols = x32.shape
    hi, lo = _pair_out(rows, cols, x32.device)
    check(_lib.load().spacer_split_f32_pair(_ptr(x32), _rowmajor(x32), _ptr(hi), _ptr(lo), cols, rows, cols, _stream()),
          "split_f32_pair")
    return hi, lo


def act_pair(x32, act, *, pre_out=None):
    """act(x) as a pair; ``pre_out`` (bf16, same shape) also receives bf16(x): the tape entry act_bwd differentiates at."""
    rows, cols = x32.shape
    hi, lo = _pair_out(rows, cols, x32.device)
    assert pre_out is None or (pre_out.dtype == BF16 and pre_out.is_contiguous() and tuple(pre_out.shape) == (rows, cols))
    check(_lib.load().spacer_act_f32_pair(_ptr(x32), _rowmajor(x32), _ptr(hi), _ptr(lo), cols, rows, cols, act, _ptr(pre_out), _stream()),
          "act_f32_pair")
    return hi, lo


def swiglu_pair(gu32, *, gu_out=None):
    """silu(gate) * up as a pair; ``gu_out`` (bf16 [rows, 2I]) also receives bf16(gate | up) for swiglu_bwd."""
    rows, two_i = gu32.shape
    assert gu32.is_contiguous() and gu32.dtype == torch.float32
    assert gu_out is None or (gu_out.dtype == BF16 and gu_out.is_contiguous() and tuple(gu_out.shape) == (rows, two_i))
    hi, lo = _pair_out(rows, two_i // 2, gu32.device)
    check(_lib.load().spacer_swiglu_f32_pair(_ptr(gu32), _ptr(hi), _ptr(lo), rows, two_i // 2, _ptr(gu_out), _stream()), "swiglu_f32_pair")
    return hi, lo


def norm_pair(x32, w, b=None, eps=1e-6, *, mean=None, rstd=None):
    """RMSNorm (b is None) or LayerNorm of fp32 rows with the output as a (hi, lo) pair; ``mean`` / ``rstd`` (fp32 [rows]) receive
    the row statistics the fast path's norm backward kernels take."""
    rows, cols = x32.shape
    assert x32.is_contiguous() and x32.dtype == torch.float32
    hi, lo = _pair_out(rows, cols, x32.device)
    check(_lib.load().spacer_norm_f32_pair(_ptr(x32), _ptr(w), _ptr(b), _ptr(hi), _ptr(lo), rows, cols, eps, int(b is not None),
                                           _ptr(mean), _ptr(rstd), _stream()), "norm_f32_pair")
    return hi, lo


def rope_pair(x32, cos, sin, rot_heads, heads, head_dim):
    """fp32 [tokens, heads*head_dim] -> rotary on the first rot_heads heads, every head split into a (hi, lo) pair."""
    tokens = x32.shape[0]
    assert x32.dtype == torch.float32 and x32.shape[1] == heads * head_dim
    assert cos.dtype == torch.float32 and cos.shape == (tokens, head_dim) and cos.is_contiguous() and sin.is_contiguous()
    hi, lo = _pair_out(tokens, heads * head_dim, x32.device)
    check(_lib.load().spacer_rope_f32_pair(_ptr(x32), _rowmajor(x32), _ptr(cos), _ptr(sin), _ptr(hi), _ptr(lo), heads * head_dim,
                                           tokens, rot_heads, heads, head_dim, _stream()), "rope_f32_pair")
    return hi, lo


def embed_fwd_f32video(ids, table, video32, video_row_of_token):
    T, H = ids.shape[0], table.shape[1]
    out = torch.empty(T, H, device=table.device, dtype=torch.float32)
    check(_lib.load().spacer_embed_fwd_f32video(_ptr(ids), _ptr(table), _ptr(video32), _ptr(video_row_of_token), _ptr(out), T, H,
                                                _stream()), "embed_fwd_f32video")
    return out


# pair attention kernel: "reg" = register-staged (round 3), "dma" = DMA-staged 256-row workgroups (round 5); default = per shape, as
# measured (scripts/probes/attn_pair_time.py): the DMA form wins on the vision tower's long frames (head_dim 80: 302 vs 378 us at
# 16 x 1024), the register-staged form on the decoder's layouts (head_dim 128: 785 vs 826 us on two cfg3 groups); same bits
ATTN_PAIR_VARIANT = {"reg": 1, "dma": 0}.get(os.environ.get("SPACER_ATTN_PAIR", ""), None)


def attn_fwd_pair(q, k, v, segs, max_q_len, Hq, Hkv, D, causal, scale, *, lse=None, variant=None):
    """attn_fwd on pair operands: q, k, v are (hi, lo) tuples of views with equal strides; returns the (hi, lo) pair of O.
    ``lse`` fp32 [Hq, T] receives the log-sum-exp rows attn_bwd reads (taped precise forward)."""
    (qh, ql), (kh, kl), (vh, vl) = q, k, v
    T = qh.shape[0]
    oh, ol = _pair_out(T, Hq * D, qh.device)
    assert qh.stride(0) == ql.stride(0) and kh.stride(0) == kl.stride(0) == vh.stride(0) == vl.stride(0)
    assert lse is None or (lse.dtype == torch.float32 and lse.is_contiguous() and tuple(lse.shape) == (Hq, T))
    check(_lib.load().spacer_attn_fwd_pair(_ptr(qh), _ptr(ql), _ptr(kh), _ptr(kl), _ptr(vh), _ptr(vl), _ptr(oh), _ptr(ol), _ptr(lse),
                                           qh.stride(0), kh.stride(0), oh.stride(0), _ptr(segs), segs.shape[0], max_q_len, T, Hq, Hkv,
                                           D, int(causal), scale,
                                           int(variant) if variant is not None else ATTN_PAIR_VARIANT if ATTN_PAIR_VARIANT is not None else int(D != 80),
                                           _stream()), "attn_fwd_pair")
    return oh, ol


# ----------------------------------------------------------------------------------------- loss
def logprob_fwd(logits32, targets, *, logp=None, lse=None):
    rows, vocab = logits32.shape
    if logp is None:
        logp = torch.empty(rows, device=logits32.device, dtype=torch.float32)
    if lse is None:
        lse = torch.empty(rows, device=logits32.device, dtype=torch.float32)
    check(_lib.load().spacer_logprob_fwd(_ptr(logits32), _rowmajor(logits32), _ptr(targets), _ptr(logp), _ptr(lse), rows,
                                         vocab, _stream()), "logprob_fwd")
    return logp, lse


def logprob_bwd(logits32, targets, lse, g, *, out=None):
    rows, vocab = logits32.shape
    if out is None:
        out = torch.empty(rows, vocab, device=logits32.device, dtype=BF16)
    check(_lib.load().spacer_logprob_bwd(_ptr(logits32), _rowmajor(logits32), _ptr(targets), _ptr(lse), _ptr(g), _ptr(out),
                                         _rowmajor(out), rows, vocab, _stream()), "logprob_bwd")
    return out


def lse_chunk_(logits32, targets, col0, state, first):
    """Running (max, sum-exp, target logit) of every row (state fp32 [3, rows]) updated with the vocabulary chunk
    [col0, col0 + cols) whose fp32 logits are ``logits32`` [rows, cols] (a view with any row stride)."""
    rows, cols = logits32.shape
    check(_lib.load().spacer_lse_chunk(_ptr(logits32), _rowmajor(logits32), _ptr(targets), col0, cols, _ptr(state[0]), _ptr(state[1]),
                                       _ptr(state[2]), rows, int(first), _stream()), "lse_chunk")


def lse_finish(state):
    rows = state.shape[1]
    logp = torch.empty(rows, device=state.device, dtype=torch.float32)
    lse = torch.empty(rows, device=state.device, dtype=torch.float32)
    check(_lib.load().spacer_lse_finish(_ptr(state[0]), _ptr(state[1]), _ptr(state[2]), _ptr(logp), _ptr(lse), rows, _stream()),
          "lse_finish")
    return logp, lse


def logprob_bwd_chunk(logits32, targets, col0, lse, g, out):
    rows, cols = logits32.shape
    check(_lib.load().spacer_logprob_bwd_chunk(_ptr(logits32), _rowmajor(logits32), _ptr(targets), col0, _ptr(lse), _ptr(g), _ptr(out),
                                               _rowmajor(out), rows, cols, _stream()), "logprob_bwd_chunk")
    return out


def grpo_loss(logp, ref_logp, adv, mask, beta):
    G, Cc = logp.shape
    dev = logp.device
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    kl = torch.empty(1, device=dev, dtype=torch.float32)
    dlogp = torch.empty(G, Cc, device=dev, dtype=torch.float32)
    check(_lib.load().spacer_grpo_loss(_ptr(logp), _ptr(ref_logp), _ptr(adv), _ptr(mask), beta, _ptr(loss), _ptr(kl),
                                       _ptr(dlogp), G, Cc, _stream()), "grpo_loss")
    return loss, kl, dlogp


def completion_mask(ids, eos_id):
    G, Cc = ids.shape
    mask = torch.empty(G, Cc, device=ids.device, dtype=torch.int32)
    lens = torch.empty(G, device=ids.device, dtype=torch.int32)
    check(_lib.load().spacer_completion_mask(_ptr(ids), eos_id, _ptr(mask), _ptr(lens), G, Cc, _stream()), "completion_mask")
    return mask, lens


def gather_f32(src, idx64, *, out=None):
    """out[j] = src.flat[idx64[j]] (fp32): d loss / d logp of the [G, C] rectangle -> the packed (EOS-trimmed) rows."""
    n = idx64.numel()
    assert src.dtype == torch.float32 and src.is_contiguous() and idx64.dtype == torch.int64
    if out is None:
        out = torch.empty(n, device=src.device, dtype=torch.float32)
    check(_lib.load().spacer_gather_f32(_ptr(src), _ptr(idx64), _ptr(out), n, _stream()), "gather_f32")
    return out


def scatter_f32_(src, idx64, dst):
    """dst.flat[idx64[j]] = src[j] (fp32): packed log-probs -> the [G, C] rectangle (zeroed by the caller)."""
    n = idx64.numel()
    assert src.dtype == torch.float32 and dst.dtype == torch.float32 and dst.is_contiguous() and src.is_contiguous() and src.numel() == n
    check(_lib.load().spacer_scatter_f32(_ptr(src), _ptr(idx64), _ptr(dst), n, _stream()), "scatter_f32")
    return dst


# ----------------------------------------------------------------------------------------- decode helpers
def eos_schedule_(logits32, step_dev, step_bias, eos_at, eos_id):
    """Synthetic completion lengths: row b's EOS logit -> -inf, or dominant when *step_dev + step_bias == eos_at[b]."""
    B, vocab = logits32.shape
    assert eos_at.dtype == torch.int32 and eos_at.numel() == B
    check(_lib.load().spacer_eos_schedule(_ptr(logits32), _rowmajor(logits32), B, vocab, _ptr(step_dev), step_bias, _ptr(eos_at), eos_id,
                                          _stream()), "eos_schedule")
    return logits32


def sample_workspace(B, vocab, device) -> torch.Tensor:
    """Scratch of the sampler's wide form (several workgroups per row for the two passes over the logits)."""
    return torch.empty(_lib.load().spacer_sample_workspace_bytes(B, vocab) // 4, device=device, dtype=torch.float32)


def sample_top_p(logits32, step_dev, *, top_k=50, top_p=0.95, temperature=1.0, seed=0, eos_id=-1, pad_id=0,
                 suppress_eos=False, finished=None, out_ids=None, out_logp=None, workspace=None):
    B, vocab = logits32.shape
    if out_ids is None:
        out_ids = torch.empty(B, device=logits32.device, dtype=torch.int64)
    check(_lib.load().spacer_sample_top_p(_ptr(logits32), _rowmajor(logits32), B, vocab, top_k, top_p, temperature, seed,
                                          _ptr(step_dev), eos_id, pad_id, int(suppress_eos), _ptr(finished), _ptr(out_ids),
                                          _ptr(out_logp), _ptr(workspace), 0 if workspace is None else workspace.numel() * 4, _stream()),
          "sample_top_p")
    return out_ids


def sample_top_p_step(logits32, step_dev, step_bias, out_matrix, *, top_k=50, top_p=0.95, temperature=1.0, seed=0, eos_id=-1, pad_id=0,
                      suppress_eos=False, finished=None, out_ids=None, workspace=None):
    """sample_top_p for the decode loop: Philox step = *step_dev + step_bias; the token also lands in out_matrix[:, step].  ``workspace``
    (sample_workspace): the wide form."""
    B, vocab = logits32.shape
    assert out_matrix.dtype == torch.int64 and out_matrix.stride(1) == 1
    if workspace is not None:
        check(_lib.load().spacer_sample_top_p_step_ws(_ptr(logits32), _rowmajor(logits32), B, vocab, top_k, top_p, temperature, seed,
                                                      _ptr(step_dev), step_bias, eos_id, pad_id, int(suppress_eos), _ptr(finished),
                                                      _ptr(out_ids), _ptr(out_matrix), out_matrix.stride(0), _ptr(workspace),
                                                      workspace.numel() * 4, _stream()), "sample_top_p_step_ws")
        return out_ids
    check(_lib.load().spacer_sample_top_p_step(_ptr(logits32), _rowmajor(logits32), B, vocab, top_k, top_p, temperature, seed,
                                               _ptr(step_dev), step_bias, eos_id, pad_id, int(suppress_eos), _ptr(finished),
                                               _ptr(out_ids), _ptr(out_matrix), out_matrix.stride(0), _stream()), "sample_top_p_step")
    return out_ids


def decode_embed(ids, table, out, counter0, counter1=None):
    """First launch of a decode step: out[b] = table[ids[b]] (fp32) and *counter0 += 1 (*counter1 += 1)."""
    B, H = ids.shape[0], table.shape[1]
    check(_lib.load().spacer_decode_embed(_ptr(ids), _ptr(table), _ptr(out), B, H, _ptr(counter0), _ptr(counter1), _stream()),
          "decode_embed")
    return out


def decode_rope_table(pos_base, step_dev, theta, cos, sin):
    B, D = cos.shape
    check(_lib.load().spacer_decode_rope_table(_ptr(pos_base), _ptr(step_dev), theta, _ptr(cos), _ptr(sin), B, D, _stream()),
          "decode_rope_table")


def decode_qkv_finish(acc32, bias, cos, sin, q_out, tail_k, tail_v, tail_len_dev, Hq, Hkv, D):
    B = acc32.shape[0]
    check(_lib.load().spacer_decode_qkv_finish(_ptr(acc32), _ptr(bias), _ptr(cos), _ptr(sin), _ptr(q_out), _ptr(tail_k),
                                               _ptr(tail_v), _ptr(tail_len_dev), B, Hq, Hkv, D, tail_k.shape[1], _stream()),
          "decode_qkv_finish")


def decode_qkv_finish_normed(acc32, bias, cos, sin, q_out, tail_k, tail_v, tail_len_dev, rowss, rowss_zero, norm_cols, eps, Hq, Hkv, D):
    """decode_qkv_finish behind gemm_skinny_packed_normed: sums scaled by rsqrt(rowss / norm_cols + eps) first; clears rowss_zero."""
    B = acc32.shape[0]
    check(_lib.load().spacer_decode_qkv_finish_normed(_ptr(acc32), _ptr(bias), _ptr(cos), _ptr(sin), _ptr(q_out), _ptr(tail_k),
                                                      _ptr(tail_v), _ptr(tail_len_dev), _ptr(rowss), _ptr(rowss_zero), norm_cols, eps,
                                                      B, Hq, Hkv, D, tail_k.shape[1], _stream()), "decode_qkv_finish_normed")


def swiglu_f32_fwd(acc32, out):
    B, two_i = acc32.shape
    check(_lib.load().spacer_swiglu_f32_fwd(_ptr(acc32), _ptr(out), B, two_i // 2, _stream()), "swiglu_f32_fwd")
    return out


# ----------------------------------------------------------------------------------------- optimizer
def sumsq_(g32, acc):
    check(_lib.load().spacer_sumsq_f32(_ptr(g32), g32.numel(), _ptr(acc), _stream()), "sumsq_f32")


def adamw_step_(master, shadow, m, v, grad, *, lr, beta1, beta2, eps, weight_decay, step, sumsq=None, max_norm=0.0,
                grad_scale=1.0):
    bc1, bc2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
    check(_lib.load().spacer_adamw_step(_ptr(master), _ptr(shadow), _ptr(m), _ptr(v), _ptr(grad), master.numel(), lr, beta1,
                                        beta2, eps, weight_decay, bc1, bc2, _ptr(sumsq), max_norm, grad_scale, _stream()),
          "adamw_step")
