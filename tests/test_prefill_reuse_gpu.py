"""GPU: the policy's prompt-side forward taken from the rollout's prefill.  The reference runs the prompt through
the model twice per step -- inside ``generate`` (SG_RLVR_trainer.py:463) and again in the scoring forward (:517-541); here
``RolloutEngine.generate`` can keep its prefill's tape (ViT + prompt rows of every decoder layer) and ``score_groups(prefill=...)``
computes the completion rows only.  Property tested: with the GEMM's K-split tail off, log-probs AND every taped tensor equal the
recompute path's bit for bit (so the backward sees identical inputs), gradients agree within the backward's own atomics noise; a
tape of other weights is refused."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny                       # noqa: E402
from spacer_amd import kernels as K                      # noqa: E402
from spacer_amd.grpo import GRPOEngine, GRPOHyper, group_advantages   # noqa: E402
from spacer_amd.qwen2vl.config import QWEN2_VL_2B, TINY  # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine      # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict, random_init_   # noqa: E402
from spacer_amd.rollout import PromptInput, RolloutEngine, SamplingParams   # noqa: E402
from spacer_amd.synthetic import make_prompt             # noqa: E402


def _flatten(t, prefix=""):
    """(name, tensor) for every tensor of a (nested) tape."""
    if isinstance(t, torch.Tensor):
        yield prefix, t
    elif isinstance(t, dict):
        for k, v in t.items():
            yield from _flatten(v, f"{prefix}.{k}")
    elif isinstance(t, (list, tuple)):
        for i, v in enumerate(t):
            yield from _flatten(v, f"{prefix}[{i}]")


def _compare(eng, prompts, comps):
    entries = [(p.ids, p.pix, p.grids) for p in prompts]
    with K.plan(gemm_no_split=1, gemm_tile=256):
        ta, tb = {}, {}
        lp_a = eng.score_groups(entries, comps, tape=ta, prefill=[p.prefill for p in prompts])
        lp_b = eng.score_groups(entries, comps, tape=tb)
    assert ta["reused_prefill"] and not tb["reused_prefill"]
    assert torch.equal(lp_a, lp_b), float((lp_a - lp_b).abs().max())
    fa, fb = dict(_flatten(ta)), dict(_flatten(tb))
    assert set(fa) == set(fb), set(fa) ^ set(fb)
    for name in fa:
        if fa[name].dtype in (torch.bfloat16, torch.float32) and ".segs" not in name:
            assert torch.equal(fa[name], fb[name]), (name, float((fa[name].float() - fb[name].float()).abs().max()))
    return lp_a, ta, tb


def test_tiny_two_groups_logps_tapes_and_gradients(dev):
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    eng = Qwen2VLEngine(TINY, params)
    roll = RolloutEngine(eng)
    roll.keep_prefill_tape = True
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    pix2, grid2 = K.patchify(g["frames"].flip(0).contiguous().to(dev), kpad=TINY.patch_kpad)
    prompts = [PromptInput(g["prompt"].to(dev), pix, [tuple(grid)]), PromptInput(g["prompt"].to(dev), pix2, [tuple(grid2)])]
    with K.plan(gemm_no_split=1, gemm_tile=256):      # the prefill whose tape is kept runs under the same launch plan as the passes compared below
        out = roll.generate(prompts, 3, SamplingParams(max_new_tokens=8, seed=3, suppress_eos=True))
    assert all(p.prefill is not None for p in prompts) and prompts[1].prefill.row0 == prompts[0].ids.numel()
    comps = [out[:3], out[3:]]
    lp, ta, tb = _compare(eng, prompts, comps)
    # one group alone (the second of the pass: a row offset into the kept tape)
    with K.plan(gemm_no_split=1, gemm_tile=256):
        t1 = {}
        lp1 = eng.score_groups([(prompts[1].ids, prompts[1].pix, prompts[1].grids)], [comps[1]], tape=t1, prefill=[prompts[1].prefill])
    assert t1["reused_prefill"] and torch.equal(lp1, lp[3:])
    # identical tapes -> the backward differs only by its atomics' arrival order: compare with the spread of the recompute path itself
    dlogp = torch.randn(6, 8, generator=torch.Generator().manual_seed(5)).to(dev) * 0.5
    grads = []
    for tape in (ta, tb):
        G = eng.W.like(torch.float32)
        eng.backward_group(tape, dlogp, G)
        grads.append(G.flat.clone())
    scale = float(grads[1].abs().max())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * scale + 1e-7
    # a tape of other weights is refused (the optimizer bumps the version)
    eng.weights_version += 1
    t2 = {}
    eng.score_groups([(p.ids, p.pix, p.grids) for p in prompts], comps, tape=t2, prefill=[p.prefill for p in prompts])
    assert not t2["reused_prefill"]


def test_cfg2_shapes_at_2b_width(dev):
    """BASELINE configs[1] shapes: Qwen2-VL-2B (depth cut to 4 + 4 layers to keep the test short), 8 frames 280x364, two groups of
    K = 4 x 64 completion tokens in one pass: the big-tile kernels, the K-concatenated rows, a 882-row prompt per group."""
    import dataclasses
    cfg = dataclasses.replace(QWEN2_VL_2B, layers=4, vit_depth=4)
    params = FlatParams.empty(cfg, dev)
    random_init_(params, seed=1234)
    eng = Qwen2VLEngine(cfg, params)
    roll = RolloutEngine(eng)
    roll.keep_prefill_tape = True
    prompts = [make_prompt(cfg, gi, 8, 280, 364, 360, dev)[0] for gi in range(3)]
    with K.plan(gemm_no_split=1, gemm_tile=256):      # (one fp32 summation order for the prefill and for both scoring passes)
        out = roll.generate(prompts, 4, SamplingParams(max_new_tokens=64, seed=9, suppress_eos=True))
    comps = [out[4 * gi:4 * gi + 4] for gi in range(3)]
    _compare(eng, prompts[1:], comps[1:])            # groups 1, 2 of a three-prompt prefill: offsets into the kept tape
    del eng, params
    torch.cuda.empty_cache()


def test_step_with_and_without_reuse(dev):
    """GRPOEngine: the same step (same rollouts, advantages) with the kept tape and with the recompute path: same loss / KL / log-probs."""
    g = load_tiny()
    res = {}
    for reuse in (True, False):
        params = FlatParams.empty(TINY, dev)
        load_state_dict(params, g["w"])
        ge = GRPOEngine(TINY, params, GRPOHyper(num_generations=3, learning_rate=1e-4, reuse_prefill=reuse))
        pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
        prompts = [PromptInput(g["prompt"].to(dev), pix, [tuple(grid)]) for _ in range(2)]
        with K.plan(skinny_blocks=1, gemm_no_split=1, gemm_tile=256):
            comp = ge.rollout(prompts, SamplingParams(max_new_tokens=8, seed=1, suppress_eos=True))
            assert (prompts[0].prefill is not None) == reuse
            adv, _ = group_advantages(torch.tensor([2.0, 0.0, 1.0]), 3)
            out = ge.score_and_backward_multi(prompts, [comp[:3], comp[3:]], [adv, adv], grad_scale=0.5)
        assert prompts[0].prefill is None
        ge.optimizer_step()
        res[reuse] = (comp.clone(), out["logps"].clone(), float(out["loss"]), float(out["kl"]), ge.policy.flat.clone())
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    assert res[True][2] == res[False][2] and res[True][3] == res[False][3]
    assert float((res[True][4].float() - res[False][4].float()).abs().max()) <= 1e-6


def test_partly_kept_tape_whole_passes_from_the_front(dev, monkeypatch):
    """Round 6: when the whole prefill tape does not fit, the first passes' prompts keep theirs (RolloutEngine._tape_keep_count) -- the
    batch runs as a taped and an untaped prefill pass into ONE prompt-KV cache.  (i) the decision: whole passes from the front, never a
    proper subset that is not a multiple of the pass size, all prompts when the whole tape is small; (ii) the rollouts equal the one-pass
    prefill's (same greedy tokens), only the first n prompts carry a slice; (iii) a scoring pass over kept prompts reuses the tape with
    the recompute pass's log-probs, a pass over the others recomputes."""
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    eng = Qwen2VLEngine(TINY, params)
    roll = RolloutEngine(eng)
    pixs = [K.patchify(g["frames"].roll(i, 0).contiguous().to(dev), kpad=TINY.patch_kpad) for i in range(5)]
    def mk():
        return [PromptInput(g["prompt"].to(dev), pixs[i][0], [tuple(pixs[i][1])]) for i in range(5)]       # same text, other frames
    # (i) the static rule on a pretend device: sizes from the tiny model, "memory" chosen around them
    prompts = mk()
    roll.keep_prefill_tape, roll.prefill_pass_size = None, 2
    class Props:
        total_memory = 0
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda d: Props)
    assert roll._tape_keep_count(prompts) == 0                       # (no memory: nothing is kept)
    Props.total_memory = 10 ** 15
    assert roll._tape_keep_count(prompts) == 5
    one = roll.prefill_tape_bytes / 5.0                              # bytes per prompt
    # (never 1 or 3: whole passes of 2)
    for mem, want in ((40, 5), (25, 4), (14, 2), (8, 0)):            # (25: 2 + (2 + 1) + 2 <= 25; 14: 4 >= 0.2 x 14; 8: 2 >= 0.2 x 8)
        roll._fit_logged.clear()
        Props.total_memory = int(mem * one)
        assert roll._tape_keep_count(prompts) == want, (mem, want)
    # whole tape: 4 x 5 < total and 5 < 0.15 total -> total > 33.4 prompts' worth.  Else whole passes of 2 from the front while
    # static + largest pass (2) + half the first pass's share (1) + the idle shares + 0.07 total <= total and the kept tape < 0.2 total
    roll.static_bytes = int(20 * one)                                # the training state beside the tape
    Props.total_memory = int(24.5 * one)
    assert roll._tape_keep_count(prompts) == 0                       # (20 + largest pass 2 + half a share 1 + 0.07 x 24.5 > 24.5: nothing)
    Props.total_memory = int(26 * one)
    assert roll._tape_keep_count(prompts) == 2                       # (20 + 2 + 1 + 1.82 <= 26, but + 2 more idle does not: the first pass only)
    roll.static_bytes = 0
    roll._fit_logged.clear()
    assert roll._tape_keep_count(prompts, [2] * 5, 0) == 4 and roll._tape_keep_count(prompts, [2] * 5, 10 ** 7) == 0   # long rollouts: larger passes
    monkeypatch.undo()
    # (ii) + (iii) with the count forced to 2 of 5
    sp = SamplingParams(max_new_tokens=5, top_k=1, top_p=1.0, suppress_eos=True)
    with K.plan(gemm_no_split=1, gemm_tile=256, skinny_blocks=1):
        roll.keep_prefill_tape = False
        base_prompts = mk()
        base = roll.generate(base_prompts, 2, sp, use_graph=False)
        roll.keep_prefill_tape = None
        monkeypatch.setattr(roll, "_tape_keep_count", lambda ps, counts=None, C=0: 2)
        prompts = mk()
        out = roll.generate(prompts, 2, sp, use_graph=False)
        assert [p.prefill is not None for p in prompts] == [True, True, False, False, False]
        assert prompts[0].prefill.shared is prompts[1].prefill.shared and prompts[1].prefill.index == 1
        assert torch.equal(out, base)
        comps = [out[2 * i:2 * i + 2] for i in range(5)]
        _compare(eng, prompts[:2], comps[:2])
        entries = [(p.ids, p.pix, p.grids) for p in prompts[2:4]]
        t = {}
        eng.score_groups(entries, comps[2:4], tape=t, prefill=None)
        assert not t["reused_prefill"]

