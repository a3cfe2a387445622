"""Run every GEMM shape of the cfg3 step (shape, launches/step from `bench.py --gemm-shapes`) a few times each, so that a
`rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE` pass over THIS script measures the HBM traffic of the dominant kernel per shape
(rocprofv3 crashes when wrapped around the whole 7B step).  Operands rotate through > 256 MB of buffers so the
Infinity Cache does not hide re-reads.   usage: python scripts/gemm_step_shapes.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacer_amd import kernels as K  # noqa: E402

# (M, N, K, launches per cfg3 step, fp32 output?)
SHAPES = [
    (5498, 37888, 3584, 448, 0), (37888, 3584, 5504, 224, 1), (5498, 3584, 18944, 448, 1), (5498, 3584, 37888, 224, 0),
    (3584, 18944, 5504, 224, 1), (5498, 18944, 3584, 224, 0), (5498, 3584, 3584, 672, 1), (4160, 5120, 1280, 768, 0),
    (5498, 4608, 3584, 448, 0), (11216, 37888, 3584, 28, 0), (4096, 152064, 3584, 16, 1), (4160, 1280, 5120, 768, 1),
    (5498, 3584, 4608, 224, 0), (4608, 3584, 5504, 224, 1), (3584, 3584, 5504, 224, 1), (4160, 1280, 1280, 768, 1),
    (4160, 3840, 1280, 512, 0), (152064, 3584, 4096, 8, 1), (11216, 3584, 18944, 28, 1), (4096, 3584, 152064, 8, 0),
    (1280, 5120, 4160, 256, 1), (5120, 1280, 4160, 256, 1), (33280, 5120, 1280, 32, 0), (3840, 1280, 4160, 256, 1),
]

if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    dev = torch.device("cuda:0")
    for M, N, Kd, _, f32 in SHAPES:
        nbuf = max(2, int(6e8 // ((M + N) * Kd * 2)) + 1)
        a = [(torch.randn(M, Kd, device=dev) * 0.1).to(torch.bfloat16) for _ in range(min(nbuf, 4))]
        b = [(torch.randn(N, Kd, device=dev) * 0.1).to(torch.bfloat16) for _ in range(min(nbuf, 4))]
        out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
        for r in range(reps):
            if (M, N, Kd) in ((5498, 37888, 3584), (11216, 37888, 3584)):
                # the gate|up projection runs with the SwiGLU in its epilogue; gate|up themselves are written on the policy pass only
                K.gemm_swiglu(a[r % len(a)], b[r % len(b)], keep_gu=(r % 2 == 0) and M == 5498)
            else:
                K.gemm_nt(a[r % len(a)], b[r % len(b)], out=out)
        torch.cuda.synchronize()
        del a, b, out
