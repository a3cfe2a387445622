"""Run every 256-tile GEMM shape of the cfg3 step (shape and launches/step from `bench.py --gemm-shapes`) a few times each, so that
a `rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` pass over THIS script measures the HBM-side traffic of the GEMM per shape (rocprofv3
--pmc around the whole 7B step crashes).  Operands rotate through > 256 MB of buffers so the Infinity Cache does not hide re-reads.
kinds: nt = forward (A[M,K] . B[N,K]^T), swiglu = gate|up with the SwiGLU epilogue, dx = dY . W in place (trans_b),
dw = dY^T . X in place with the fp32 accumulate epilogue (trans_a + trans_b).     usage: python scripts/gemm_step_shapes.py [reps] [kind ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (kind, M, N, K, launches per cfg3 step, fp32 output?, accumulate into the output (fp32 residual == C)?) -- the step as `bench.py`
# runs it since round 2: two prompt groups per scoring pass (T = 10 996 rows, 4 passes per step for the reference and the policy
# each), the lm_head in vocabulary chunks of 8192 rows, ViT over two groups' patches (8320 rows); counts from `bench.py --gemm-shapes`
SHAPES = [
    ("swiglu", 10996, 37888, 3584, 224, 0, 0), ("nt", 10996, 3584, 18944, 224, 1, 1), ("nt", 10996, 4608, 3584, 224, 0, 0),
    ("nt", 10996, 3584, 3584, 224, 1, 1), ("nt", 8192, 8192, 3584, 144, 1, 0),
    ("swiglu", 11216, 37888, 3584, 28, 0, 0), ("nt", 11216, 3584, 18944, 28, 1, 1), ("nt", 11216, 4608, 3584, 28, 0, 0), ("nt", 11216, 3584, 3584, 28, 1, 1),
    ("nt", 8320, 5120, 1280, 256, 0, 0), ("nt", 8320, 3840, 1280, 256, 0, 0), ("nt", 8320, 1280, 5120, 256, 1, 1),
    ("dx", 10996, 3584, 37888, 112, 0, 0), ("dx", 10996, 18944, 3584, 112, 0, 0), ("dx", 10996, 3584, 4608, 112, 0, 0), ("dx", 10996, 3584, 3584, 112, 0, 0),
    ("dx", 8192, 3584, 8192, 72, 1, 1), ("dx", 8320, 5120, 1280, 128, 0, 0), ("dx", 8320, 1280, 5120, 128, 0, 0), ("dx", 8320, 1280, 3840, 128, 0, 0),
    ("dw", 37888, 3584, 10996, 112, 1, 1), ("dw", 3584, 18944, 10996, 112, 1, 1), ("dw", 4608, 3584, 10996, 112, 1, 1), ("dw", 3584, 3584, 10996, 112, 1, 1),
    ("dw", 8192, 3584, 8192, 72, 1, 1), ("dw", 1280, 5120, 8320, 128, 1, 1), ("dw", 5120, 1280, 8320, 128, 1, 1), ("dw", 3840, 1280, 8320, 128, 1, 1),
]


def run(reps: int, kinds=None):
    from spacer_amd import kernels as K
    dev = torch.device("cuda:0")
    for kind, M, N, Kd, _, f32, acc in SHAPES:
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
                K.gemm_swiglu(x, y, keep_gu=(r % 2 == 0) and M == 10996)
            elif kind == "nt":
                K.gemm_nt(x, y, out=out, residual=out if acc else None)
            elif kind == "dx":
                K.gemm(x, y, trans_b=True, out=out, residual=out if acc else None)
            else:
                K.gemm(x, y, trans_a=True, trans_b=True, out=out, residual=out)
        torch.cuda.synchronize()
        del a, b, out


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 2, set(sys.argv[2:]) or None)
