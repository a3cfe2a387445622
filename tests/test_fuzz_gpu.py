"""GPU: a seeded 45-second slice of the randomised kernel sweep (scripts/fuzz_kernels.py): ragged GEMM shapes in every epilogue /
operand mode (incl. the in-place dX / dW forms and the K-split tail), SwiGLU-epilogue GEMMs bit-compared with the two-step path,
attention forward / backward on random segment layouts, decode attention (shared and per-sequence), packed skinny GEMMs, norms,
and the front-end resize -- each against fp32 torch.  The full sweep (minutes, other seeds) stays a hand-run script."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def test_seeded_fuzz_slice(dev):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "fuzz_kernels.py")
    spec = importlib.util.spec_from_file_location("fuzz_kernels", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    counts = mod.run(45.0, seed=20250928)
    print("fuzz slice:", counts)
    assert sum(counts.values()) > 300 and all(v > 0 for v in counts.values()), counts
