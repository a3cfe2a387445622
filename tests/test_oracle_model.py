"""CPU: the fp32 oracle reproduces HF transformers' Qwen2-VL (golden vectors generated in the authoring
container by scripts/make_golden_model.py; HF is the third-party model the reference trainer calls).
Tolerance: fp32 round-off of a 2-layer model, 5e-5 absolute on logits/log-probs."""
import torch

from golden_util import load_tiny
from oracle import qwen2vl_fp32 as O


def test_vit_matches_hf():
    g = load_tiny()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    assert tuple(grid) == g["grid"]
    out = O.vit_forward(g["w"], g["cfg"], rows, [grid])
    assert torch.allclose(out, g["hf_vit"], atol=2e-5, rtol=1e-4)


def test_logits_and_logps_match_hf():
    g = load_tiny()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    ids = torch.cat([g["prompt"], g["completions"][0]])
    lg = O.full_logits(g["w"], g["cfg"], ids, rows, [grid])
    assert (lg - g["hf_logits_row0"]).abs().max() < 5e-5
    lp = O.completion_logps(g["w"], g["cfg"], g["prompt"], g["completions"], rows, [grid])
    assert (lp - g["hf_logps"]).abs().max() < 5e-5


def test_tied_embedding_fixture_matches_hf():
    """tie_word_embeddings=True (the Qwen2-VL-2B arrangement, BASELINE configs[0]/[1]): lm_head is the embedding table."""
    g = load_tiny("tiny_tied_model.npz")
    assert g["cfg"]["tie_embeddings"] and "lm_head.weight" not in g["w"]
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    ids = torch.cat([g["prompt"], g["completions"][0]])
    lg = O.full_logits(g["w"], g["cfg"], ids, rows, [grid])
    assert (lg - g["hf_logits_row0"]).abs().max() < 5e-5
    lp = O.completion_logps(g["w"], g["cfg"], g["prompt"], g["completions"], rows, [grid])
    assert (lp - g["hf_logps"]).abs().max() < 5e-5


def test_engine_emulator_is_the_oracle_without_rounding_and_budget_is_additive():
    """oracle/qwen2vl_engine_emul.py (the engine's bf16 rounding points on the CPU): with no class rounding it IS the
    oracle; with every class as a hi+lo pair it is within 1e-5; with every class rounding it sits at the bf16-operand floor
     -- above the north-star's 1e-3 even on this 2-layer model."""
    from oracle import qwen2vl_engine_emul as E
    g = load_tiny()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    w = {k: v.to(torch.bfloat16).float() for k, v in g["w"].items()}
    rows = rows.to(torch.bfloat16).float()
    gen = torch.Generator().manual_seed(0)
    comps = torch.randint(5, 990, (8, 16), generator=gen)
    want = O.completion_logps(w, g["cfg"], g["prompt"], comps, rows, [grid])
    none = E.completion_logps(w, g["cfg"], g["prompt"], comps, rows, [grid], points=())
    assert (none - want).abs().max() < 2e-6
    pairs = E.completion_logps(w, g["cfg"], g["prompt"], comps, rows, [grid], split=E.ALL_POINTS)
    assert (pairs - want).abs().max() < 1e-5
    full = E.completion_logps(w, g["cfg"], g["prompt"], comps, rows, [grid])
    rms = float((full - want).pow(2).mean().sqrt())
    assert 2e-4 < rms < 2e-3, rms


def test_mrope_positions_match_hf_and_era_rule():
    g = load_tiny()
    ids = torch.cat([g["prompt"], g["completions"][0]]).tolist()
    pos, delta = O.mrope_position_ids(ids, [g["grid"]], g["cfg"])
    assert torch.equal(pos, g["hf_pos"]) and delta == g["hf_delta"]
    # era (transformers 4.x) rule agrees when gt <= max(gh, gw)/merge, and differs for a long thin video
    pos_e, _ = O.mrope_position_ids(ids, [g["grid"]], g["cfg"], era_rule=True)
    assert torch.equal(pos_e, pos)
    cfg = g["cfg"]
    thin = [cfg["video_token_id"]] * (8 * 1 * 1) + [5, 6]
    p5, _ = O.mrope_position_ids(thin, [(8, 2, 2)], cfg)
    p4, _ = O.mrope_position_ids(thin, [(8, 2, 2)], cfg, era_rule=True)
    assert int(p5[0, -2]) == 1 and int(p4[0, -2]) == 8


def test_shared_prefix_mask_equals_independent_rows():
    """The engine's packed layout (prompt once, K rollouts attending it) is the same function as K independent
    causal rows -- checked here on the oracle itself so the GPU test can rely on either form."""
    g = load_tiny()
    cfg, w = g["cfg"], g["w"]
    rows, grid = O.patchify_frames(g["frames"], cfg)
    ve = O.vit_forward(w, cfg, rows, [grid])
    P, (Kn, C) = g["prompt"].numel(), g["completions"].shape
    ids = torch.cat([g["prompt"], g["completions"].reshape(-1)])
    e = O.embed_with_video(w, cfg, ids, ve)
    pos3, delta = O.mrope_position_ids(g["prompt"].tolist(), [grid], cfg)
    comp = (P + delta + torch.arange(C)).view(1, C).expand(3, C)
    pos = torch.cat([pos3] + [comp] * Kn, 1)
    T = P + Kn * C
    mask = torch.zeros(T, T, dtype=torch.bool)
    mask[:P, :P] = torch.ones(P, P, dtype=torch.bool).tril()
    for k in range(Kn):
        a = P + k * C
        mask[a:a + C, :P] = True
        mask[a:a + C, a:a + C] = torch.ones(C, C, dtype=torch.bool).tril()
    lg = O.llm_forward(w, cfg, e, pos, mask)
    lp = torch.log_softmax(lg, -1)
    got = torch.stack([torch.stack([lp[P - 1 if t == 0 else P + k * C + t - 1, g["completions"][k, t]] for t in range(C)])
                       for k in range(Kn)])
    assert (got - g["hf_logps"]).abs().max() < 5e-5


def test_patchify_matches_hf_processor():
    """K1's restatement against HF's own pre-processing output (tests/golden/patchify_hf.npz, scripts/make_golden_patchify.py):
    rescale, CLIP normalisation, temporal pairing and merge-block-major patch order, bit for bit."""
    import numpy as np
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "patchify_hf.npz"))
    cfg = dict(patch=14, tpatch=2, merge=2)
    for i in range(3):
        rows, grid = O.patchify_frames(torch.from_numpy(z[f"frames{i}"]), cfg)
        assert tuple(grid) == tuple(int(v) for v in z[f"grid{i}"])
        assert torch.equal(rows, torch.from_numpy(z[f"pixel_values{i}"]))
