"""CPU: oracle/cpu_path.py -- the whole SG-RLVR step for one prompt group restated on the host (BASELINE.json configs[0]:
"CPU eager reference (plumbing, no GPU)"; also bench.py's cpu_baseline).  Checked against the pinned oracle pieces:
shared-prompt scoring == K independent rows (HF-pinned numbers), KV-cache decode == teacher-forced forward, and one full
step (rollout -> masks -> ref/policy log-probs -> GRPO loss -> autograd backward) at cfg1's plumbing shape."""
import torch

from golden_util import load_tiny
from oracle import cpu_path as CP
from oracle import qwen2vl_fp32 as O


def _case():
    g = load_tiny()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    return g, rows, [tuple(grid)]


def test_group_logps_equal_hf_pinned_rows():
    g, rows, grids = _case()
    lp = CP.group_logps(g["w"], g["cfg"], g["prompt"], g["completions"], rows, grids)
    assert (lp - g["hf_logps"]).abs().max() < 5e-5


def test_kv_cache_decode_equals_teacher_forcing():
    g, rows, grids = _case()
    cfg, w = g["cfg"], g["w"]
    with torch.no_grad():
        ve = O.vit_forward(w, cfg, rows, grids)
        e = O.embed_with_video(w, cfg, g["prompt"], ve)
        pos3, delta = O.mrope_position_ids(g["prompt"].tolist(), grids, cfg)
        first, pk, pv = CP.prefill(w, cfg, e, pos3)
        comps = g["completions"]
        Kn, C = comps.shape
        P = g["prompt"].numel()
        tk = [torch.zeros(Kn, C, cfg["kv_heads"], cfg["head_dim"]) for _ in range(cfg["layers"])]
        tv = [torch.zeros(Kn, C, cfg["kv_heads"], cfg["head_dim"]) for _ in range(cfg["layers"])]
        step_logits = [first.view(1, -1).expand(Kn, -1)]
        for t in range(C - 1):
            step_logits.append(CP.decode_step(w, cfg, comps[:, t], P + delta + t, pk, pv, tk, tv, t))
        got = torch.stack(step_logits, 1)                                    # (K, C, V): logits that predict token t
        for k in range(Kn):
            full = O.full_logits(w, cfg, torch.cat([g["prompt"], comps[k]]), rows, grids)
            assert (got[k] - full[P - 1:P - 1 + C]).abs().max() < 5e-5


def test_full_cpu_step_cfg1_plumbing_shape():
    """cfg1's shape on the miniature: 2 prompts x 4 frames x K=2 rollouts, everything on the CPU."""
    g, rows, grids = _case()
    cfg = g["cfg"]
    for prompt_idx in range(2):
        w = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in g["w"].items()}
        out = CP.grpo_group_step(w, g["w"], cfg, g["prompt"], rows, grids, num_generations=2, max_new_tokens=6,
                                 eos_token_id=7, seed=prompt_idx, rewards=torch.tensor([2.0, 0.0]))
        assert out["completions"].shape == (2, 6) and int((out["completions"] == 7).sum()) == 0
        assert torch.isfinite(out["logps"]).all() and abs(out["loss"]) < 10
        # policy == reference weights at step 0: KL = 0 and the loss is -mean(A) = 0 for a normalised group
        assert (out["logps"] - out["ref_logps"]).abs().max() < 1e-5 and abs(out["loss"]) < 1e-5
        gn = sum(float(v.grad.pow(2).sum()) for v in w.values() if v.grad is not None) ** 0.5
        assert gn > 0 and gn == gn
        assert set(out["seconds"]) == {"vit+prefill", "decode", "ref scoring", "policy scoring", "loss+backward"}
