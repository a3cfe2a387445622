"""A/B of the 256-tile GEMM across BUILDS of the library in ONE process on ONE box (did round 4's kt_wrap scalar
selects in the DMA-issue path cost the NT forms 1-3 %, or was the r03 -> r04 drop box-to-box spread?).

Each argument is `label=path/to/libspacer_hip.so` (e.g. r03 / r04 builds under scripts/probes/_variants/ and the tree's own
library).  Every build runs the step's GEMM shapes (two cfg3 groups per pass, M = 10 996) through the SAME entry points with a NULL
launch plan; builds alternate per repetition (A B C A B C ...), HIP events around a run of launches.  Prints a markdown table of
TF/s per build and shape (median over the repetitions) and the spread.

    python scripts/probes/gemm_ab_libs.py r03=scripts/probes/_variants/libspacer_r03.so r04=... r05=spacer_amd/libspacer_hip.so"""
import ctypes as C
import statistics
import sys

import torch

REPS, LAUNCHES = 9, 6
_p, _i, _l = C.c_void_p, C.c_int, C.c_long


class Epi(C.Structure):          # spacer_gemm_epilogue (the plan pointer exists since round 4; NULL = defaults; r03 never reads it)
    _fields_ = [("bias", _p), ("residual", _p), ("ldr", _l), ("out_f32", _i), ("act", _i), ("alpha", C.c_float), ("workspace", _p),
                ("workspace_bytes", _l), ("plan", _p)]


def load(path):
    lib = C.CDLL(path)
    lib.spacer_gemm_bf16_nt.argtypes = [_p, _l, _p, _l, _p, _l, _i, _i, _i, C.POINTER(Epi), _p]
    lib.spacer_gemm_bf16.argtypes = [_p, _l, _p, _l, _p, _l, _i, _i, _i, _i, _i, C.POINTER(Epi), _p]
    lib.spacer_gemm_swiglu_bf16.argtypes = [_p, _l, _p, _l, _p, _p, _l, _p, _l, _i, _i, _i, _p]
    lib.spacer_gemm_workspace_bytes.restype = _l
    lib.spacer_last_error.restype = C.c_char_p
    return lib


def main():
    libs = [(a.split("=", 1)[0], load(a.split("=", 1)[1])) for a in sys.argv[1:]]
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s: (torch.randn(*s, device=dev, generator=g) * 0.05).to(torch.bfloat16)      # noqa: E731
    T, H, I, QKV = 10996, 3584, 18944, 4608
    x, xi = rnd(T, H), rnd(T, I)
    w_gu, w_qkv, w_o, w_dn = rnd(2 * I, H), rnd(QKV, H), rnd(H, H), rnd(H, I)
    dgu = rnd(T, 2 * I)
    res = torch.randn(T, H, device=dev, generator=g)
    out_gu, out_a, out_qkv = torch.empty(T, 2 * I, device=dev, dtype=torch.bfloat16), torch.empty(T, I, device=dev, dtype=torch.bfloat16), \
        torch.empty(T, QKV, device=dev, dtype=torch.bfloat16)
    out32, dx = torch.empty(T, H, device=dev), torch.empty(T, H, device=dev, dtype=torch.bfloat16)
    gw = torch.zeros(2 * I, H, device=dev)
    ws = torch.empty(libs[-1][1].spacer_gemm_workspace_bytes() // 4 + 64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    b_qkv, Tv, Hv = rnd(QKV), 13824, 1280                                                       # vision tower: 1280 wide, 3840 q|k|v, 5120 MLP
    xv, wv_qkv, wv_fc1, bv_qkv, bv_fc1 = rnd(Tv, Hv), rnd(3 * Hv, Hv), rnd(4 * Hv, Hv), rnd(3 * Hv), rnd(4 * Hv)
    outv_qkv, outv_fc1 = torch.empty(Tv, 3 * Hv, device=dev, dtype=torch.bfloat16), torch.empty(Tv, 4 * Hv, device=dev, dtype=torch.bfloat16)

    def epi(residual=None, f32=0, bias=None, act=0):
        return Epi(bias.data_ptr() if bias is not None else None, residual.data_ptr() if residual is not None else None, residual.stride(0) if residual is not None else 0, f32, act, 1.0,
                   ws.data_ptr(), ws.numel() * 4, None)

    def nt(lib, a, w, out, e):
        return lib.spacer_gemm_bf16_nt(a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), a.shape[0], w.shape[0],
                                       a.shape[1], C.byref(e), stream)

    cases = [   # name, flops, launcher
        ("gate|up + SwiGLU <t,f,f,t>", 2.0 * T * 2 * I * H,
         lambda lib: lib.spacer_gemm_swiglu_bf16(x.data_ptr(), H, w_gu.data_ptr(), H, None, out_a.data_ptr(), I, out_gu.data_ptr(), 2 * I, T, I, H, stream)),
        ("q|k|v bf16 out <t,f,f,t>", 2.0 * T * QKV * H, lambda lib: nt(lib, x, w_qkv, out_qkv, epi())),
        ("q|k|v + bias bf16 out <t,f,f,t>", 2.0 * T * QKV * H, lambda lib: nt(lib, x, w_qkv, out_qkv, epi(bias=b_qkv))),
        ("ViT q|k|v + bias (13824 x 3840 x 1280)", 2.0 * Tv * 3 * Hv * Hv, lambda lib: nt(lib, xv, wv_qkv, outv_qkv, epi(bias=bv_qkv))),
        ("ViT fc1 + bias + act (13824 x 5120 x 1280)", 2.0 * Tv * 4 * Hv * Hv, lambda lib: nt(lib, xv, wv_fc1, outv_fc1, epi(bias=bv_fc1, act=1))),
        ("o fp32 + residual <t,f,f,f>", 2.0 * T * H * H, lambda lib: nt(lib, x, w_o, out32, epi(res, 1))),
        ("down fp32 + residual <t,f,f,f>", 2.0 * T * H * I, lambda lib: nt(lib, xi, w_dn, out32, epi(res, 1))),
        ("dX of gate|up <t,f,t,t>", 2.0 * T * H * 2 * I,
         lambda lib: lib.spacer_gemm_bf16(dgu.data_ptr(), 2 * I, w_gu.data_ptr(), H, dx.data_ptr(), H, T, H, 2 * I, 0, 1, C.byref(epi()), stream)),
        ("dW of gate|up <t,t,t,f>", 2.0 * T * H * 2 * I,
         lambda lib: lib.spacer_gemm_bf16(dgu.data_ptr(), 2 * I, x.data_ptr(), H, gw.data_ptr(), H, 2 * I, H, T, 1, 1, C.byref(epi(gw, 1)), stream)),
    ]
    table = {}
    for cname, flops, fn in cases:
        for label, lib in libs:                     # warm-up + error check
            rc = fn(lib)
            assert rc == 0, (label, cname, lib.spacer_last_error())
        torch.cuda.synchronize()
        for rep in range(REPS):
            for label, lib in libs:                 # A B C A B C ...
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(LAUNCHES):
                    fn(lib)
                e1.record()
                torch.cuda.synchronize()
                table.setdefault((cname, label), []).append(flops * LAUNCHES / (e0.elapsed_time(e1) * 1e-3) / 1e12)
    labels = [l for l, _ in libs]
    print(f"| shape (T = {T}) | " + " | ".join(f"{l} TF/s (median of {REPS}, min-max)" for l in labels) + " |")
    print("|---|" + "---:|" * len(labels))
    for cname, _, _ in cases:
        cells = []
        for l in labels:
            v = table[(cname, l)]
            cells.append(f"{statistics.median(v):.0f} ({min(v):.0f}-{max(v):.0f})")
        print(f"| {cname} | " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
