"""Run every 256-tile GEMM shape of the cfg3 step (shape and launches/step from `bench.py --gemm-shapes`) a few times each, so that
a `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` pass over THIS script measures the HBM-side traffic of the GEMM per shape (rocprofv3
--pmc around the whole 7B step crashes).  Operands rotate through > 256 MB of buffers so the Infinity Cache does not hide re-reads.
kinds: nt = forward (A[M,K] . B[N,K]^T), swiglu = gate|up with the SwiGLU epilogue, dx = dY . W in place (trans_b),
dw = dY^T . X in place with the fp32 accumulate epilogue (trans_a + trans_b).     usage: python scripts/gemm_step_shapes.py [reps] [kind ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (kind, M, N, K, launches per cfg3 step, fp32 output (+ residual)?)
SHAPES = [
    ("swiglu", 5498, 37888, 3584, 448, 0), ("nt", 5498, 3584, 18944, 448, 1), ("nt", 5498, 4608, 3584, 448, 0), ("nt", 5498, 3584, 3584, 448, 1),
    ("nt", 4096, 152064, 3584, 16, 1), ("swiglu", 11216, 37888, 3584, 28, 0), ("nt", 11216, 3584, 18944, 28, 1),
    ("nt", 11216, 4608, 3584, 28, 0), ("nt", 11216, 3584, 3584, 28, 1),
    ("dx", 5498, 3584, 37888, 224, 0), ("dx", 5498, 18944, 3584, 224, 0), ("dx", 5498, 3584, 4608, 224, 0), ("dx", 5498, 3584, 3584, 224, 0),
    ("dx", 4096, 3584, 152064, 8, 0), ("dx", 4160, 5120, 1280, 256, 0), ("dx", 4160, 1280, 5120, 256, 0), ("dx", 4160, 1280, 3840, 256, 0),
    ("dw", 37888, 3584, 5498, 224, 1), ("dw", 3584, 18944, 5498, 224, 1), ("dw", 4608, 3584, 5498, 224, 1), ("dw", 3584, 3584, 5498, 224, 1),
    ("dw", 152064, 3584, 4096, 8, 1), ("dw", 1280, 5120, 4160, 256, 1), ("dw", 5120, 1280, 4160, 256, 1), ("dw", 3840, 1280, 4160, 256, 1),
]


def run(reps: int, kinds=None):
    from spacer_amd import kernels as K
    dev = torch.device("cuda:0")
    for kind, M, N, Kd, _, f32 in SHAPES:
        if kinds and kind not in kinds:
            continue
        nb = 2 if (M + N) * Kd * 2 > 3e8 else 4

        def mk(r, c):
            return [(torch.randn(r, c, device=dev) * 0.1).to(torch.bfloat16) for _ in range(nb)]
        if kind in ("nt", "swiglu"):
            a, b = mk(M, Kd), mk(N, Kd)
        elif kind == "dx":
            a, b = mk(M, Kd), mk(Kd, N)
        else:
            a, b = mk(Kd, M), mk(Kd, N)
        out = torch.zeros(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16) if kind != "swiglu" else None
        for r in range(reps):
            x, y = a[r % nb], b[r % nb]
            if kind == "swiglu":      # gate|up themselves are written on the policy pass only (1 of 2 launches)
                K.gemm_swiglu(x, y, keep_gu=(r % 2 == 0) and M == 5498)
            elif kind == "nt":
                K.gemm_nt(x, y, out=out, residual=out if f32 and N != 152064 else None)
            elif kind == "dx":
                K.gemm(x, y, trans_b=True, out=out)
            else:
                K.gemm(x, y, trans_a=True, trans_b=True, out=out, residual=out)
        torch.cuda.synchronize()
        del a, b, out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 2, set(sys.argv[2:]) or None)
