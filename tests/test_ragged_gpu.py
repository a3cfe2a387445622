"""GPU: EOS-trimmed scoring (round 6).  The reference builds its completion mask from the first EOS (SG_RLVR_trainer.py:493-498) and
multiplies it into the loss (:640-643): every position behind a rollout's first EOS contributes exactly zero to loss, KL and
gradients.  ``score_groups(lengths=...)`` therefore packs only the first lengths[i] tokens of rollout i into the scoring passes and
the backward.  Properties tested (GEMM K-split tail off = one fp32 summation order per row):
  * log-probs on the unmasked positions: BIT-identical to the rectangular [K, C] pass (hence loss and KL, which are functions of
    them: equal to the last digit of the loss kernel's atomic row sums), fast and precise mode, with and
    without the kept prefill tape, on lengths {1, 7, C/2, C (EOS last), no EOS};
  * gradients: equal to the rectangular pass within the backward's own summation-order noise (the pad rows of the rectangular pass
    carry exactly-zero gradients, but dropping them moves the other rows to different contraction slots of the dW GEMMs, and the
    shared prompt's dK / dV collect their K rollouts through fp32 atomics in arrival order: neither pass is bit-reproducible there);
  * the oracle (oracle/qwen2vl_fp32.py + oracle/grpo_ref.py) agrees with the trimmed step as it did with the rectangular one;
  * the seeded synthetic-length rollout mode (``SamplingParams.synthetic_lengths``) yields exactly the scheduled lengths, eager and
    graph-replayed."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny                       # noqa: E402
from oracle import grpo_ref as GR                        # noqa: E402
from oracle import qwen2vl_fp32 as O                     # noqa: E402
from spacer_amd import kernels as K                      # noqa: E402
from spacer_amd.grpo import GRPOEngine, GRPOHyper, group_advantages   # noqa: E402
from spacer_amd.qwen2vl.config import QWEN2_VL_2B, TINY  # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine      # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict, random_init_   # noqa: E402
from spacer_amd.rollout import PromptInput, RolloutEngine, SamplingParams   # noqa: E402
from spacer_amd.synthetic import make_prompt             # noqa: E402


def _completions(cfg, Kn, C, lengths, seed, dev):
    """[Kn, C] ids: random ordinary tokens, EOS as token lengths[k] - 1 (None: no EOS), pad behind it -- what generate returns."""
    g = torch.Generator().manual_seed(seed)
    lo, hi = (1000, 150000) if cfg.vocab > 150000 else (3, cfg.vocab - 1)
    ids = torch.randint(lo, hi, (Kn, C), generator=g)
    special = {cfg.eos_token_id, cfg.pad_token_id, cfg.video_token_id, cfg.image_token_id}
    for sp_id in special:
        ids[ids == sp_id] = lo
    for k, n in enumerate(lengths):
        if n is not None:
            ids[k, n - 1] = cfg.eos_token_id
            ids[k, n:] = cfg.pad_token_id
    return ids.to(dev)


def _tiny(dev):
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    pix2, grid2 = K.patchify(g["frames"].flip(0).contiguous().to(dev), kpad=TINY.patch_kpad)
    prompts = [PromptInput(g["prompt"].to(dev), pix, [tuple(grid)]), PromptInput(g["prompt"].to(dev), pix2, [tuple(grid2)])]
    return g, params, prompts


def _check_pass(eng, prompts, comps, *, precise=False, backward=True):
    """Trimmed vs rectangular pass of one engine: log-probs on the mask bit for bit, zeros elsewhere, gradients to summation order."""
    cfg = eng.cfg
    entries = [(p.ids, p.pix, p.grids) for p in prompts]
    mask, lengths = K.completion_mask(torch.cat(comps, 0), cfg.eos_token_id)
    lens = lengths.tolist()
    with K.plan(gemm_no_split=1, gemm_tile=256):
        ta, tb = {}, {}
        lp_t = eng.score_groups(entries, comps, tape=ta, precise=precise, lengths=lens)
        lp_r = eng.score_groups(entries, comps, tape=tb, precise=precise)
    m = mask.bool()
    assert ta["pack_idx"] is not None and tb["pack_idx"] is None
    assert ta["T"] == sum(p.ids.numel() for p in prompts) + sum(lens) < tb["T"]
    assert torch.equal(lp_t[m], lp_r[m]), float((lp_t[m] - lp_r[m]).abs().max())
    assert float(lp_t[~m].abs().max()) == 0.0 if (~m).any() else True
    if backward:
        dlogp = torch.randn(lp_r.shape, generator=torch.Generator().manual_seed(5)).to(lp_r.device) * 0.5 * mask.float()
        grads = []
        for tape in (ta, tb):
            G = eng.W.like(torch.float32)
            with K.plan(gemm_no_split=1, gemm_tile=256):
                eng.backward_group(tape, dlogp, G)
            grads.append(G)
        # The pad rows of the rectangular pass carry exactly-zero gradients; dropping them (a) moves the other rows to other
        # contraction slots of the dW GEMMs (fp32 summation order: ~1e-7) and (b) changes the arrival order of the fp32 atomics through
        # which the shared prompt's dK / dV collect their K rollouts -- a 1e-7 difference there flips bf16 roundings of d_qkv (one ulp =
        # 4e-3 of an element), which everything upstream inherits.  (b) is the backward's own run-to-run noise: the SAME rectangular pass
        # run twice differs by up to ~1e-4 .. 1e-3 relative Frobenius on single tensors (scripts/probes/ragged_grad_diff.py).  So:
        # tensors whose gradient is final BEFORE the first attention backward must agree to summation order; the rest to a few bf16 flips.
        L = cfg.layers - 1
        early = {"llm.lm_head", "llm.norm_w", f"llm.{L}.down_w", f"llm.{L}.gu_w", f"llm.{L}.ln2_w", f"llm.{L}.o_w"}
        for spec in grads[0].specs:
            a, b = grads[0][spec.name].float(), grads[1][spec.name].float()
            den = float(b.norm())
            if den > 0:
                rel = float((a - b).norm()) / den
                assert rel <= (2e-6 if spec.name in early else 5e-3), (spec.name, rel)
    return lp_t, lp_r, mask, lens


@pytest.mark.parametrize("precise", [False, True])
def test_trimmed_pass_equals_the_rectangular_pass_tiny(dev, precise):
    g, params, prompts = _tiny(dev)
    eng = Qwen2VLEngine(TINY, params)
    C = 16
    comps = [_completions(TINY, 5, C, [1, 7, C // 2, C, None], 11, dev), _completions(TINY, 5, C, [C // 2, None, 2, 1, C - 1], 12, dev)]
    lp_t, lp_r, mask, lens = _check_pass(eng, prompts, comps, precise=precise)
    assert lens == [1, 7, 8, 16, 16, 8, 16, 2, 1, 15]
    # the oracle on the same rollouts (fp32 CPU restatement of TR:353-366), unmasked positions only
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    want = O.completion_logps(wb, g["cfg"], g["prompt"], comps[0].cpu(), rows.to(torch.bfloat16).float(), [tuple(grid)])
    m0 = mask[:5].bool().cpu()
    err = float((lp_t[:5].cpu() - want)[m0].abs().max())
    assert err < (1e-4 if precise else 5e-3), err
    # one group alone, all rollouts short
    _check_pass(eng, prompts[:1], [_completions(TINY, 3, C, [2, 3, 1], 13, dev)], precise=precise)


def test_trimmed_pass_on_the_kept_prefill_tape(dev):
    """The policy pass that takes its prompt rows from the rollout's prefill tape (round 5), trimmed: same bits as the full pass."""
    _, params, prompts = _tiny(dev)
    eng = Qwen2VLEngine(TINY, params)
    roll = RolloutEngine(eng)
    roll.keep_prefill_tape = True
    C = 16
    with K.plan(gemm_no_split=1, gemm_tile=256):
        roll.generate(prompts, 3, SamplingParams(max_new_tokens=4, seed=3, suppress_eos=True))
    comps = [_completions(TINY, 3, C, [5, None, 1], 21, dev), _completions(TINY, 3, C, [C, 9, 2], 22, dev)]
    entries = [(p.ids, p.pix, p.grids) for p in prompts]
    mask, lengths = K.completion_mask(torch.cat(comps, 0), TINY.eos_token_id)
    with K.plan(gemm_no_split=1, gemm_tile=256):
        ta, tb = {}, {}
        lp_a = eng.score_groups(entries, comps, tape=ta, prefill=[p.prefill for p in prompts], lengths=lengths.tolist())
        lp_b = eng.score_groups(entries, comps, tape=tb)
    assert ta["reused_prefill"] and not tb["reused_prefill"] and ta["pack_idx"] is not None
    m = mask.bool()
    assert torch.equal(lp_a[m], lp_b[m])
    dlogp = torch.randn(6, C, generator=torch.Generator().manual_seed(5)).to(dev) * 0.5 * mask.float()
    grads = []
    for tape in (ta, tb):
        G = eng.W.like(torch.float32)
        eng.backward_group(tape, dlogp, G)
        grads.append(G.flat.clone())
    assert float((grads[0] - grads[1]).norm()) <= 2e-3 * float(grads[1].norm())       # (a few bf16 flips behind the dK / dV atomics: see _check_pass)


def test_trimmed_pass_at_2b_width(dev):
    """Qwen2-VL-2B widths (depth cut to 4 + 4), 8 frames 280x364, two groups of K = 4 x 96 tokens with EOS at {1, 7, 48, none} /
    {96, 33, 2, 64}: the 256-tile kernels, ragged M in every GEMM, attention segments of 1 ... 96 rows behind a 882-row prompt."""
    cfg = dataclasses.replace(QWEN2_VL_2B, layers=4, vit_depth=4)
    params = FlatParams.empty(cfg, dev)
    random_init_(params, seed=1234)
    eng = Qwen2VLEngine(cfg, params)
    prompts = [make_prompt(cfg, gi, 8, 280, 364, 360, dev)[0] for gi in range(2)]
    C = 96
    comps = [_completions(cfg, 4, C, [1, 7, C // 2, None], 31, dev), _completions(cfg, 4, C, [C, 33, 2, 64], 32, dev)]
    _check_pass(eng, prompts, comps)
    _check_pass(eng, prompts, comps, precise=True, backward=False)
    del eng, params
    torch.cuda.empty_cache()


def test_step_trimmed_equals_rectangular_and_the_oracle(dev):
    """GRPOEngine.score_and_backward_multi with trim_completions on / off on the same rollouts (reference model != policy so that the
    KL term is live): same log-probs on the mask bit for bit, same loss and KL (to the loss kernel's atomics), same updated weights; loss / KL / mask agree with
    oracle/grpo_ref.py evaluated on the oracle's own log-probs as before."""
    g, _, _ = _tiny(dev)
    C = 16
    res = {}
    for trim in (True, False):
        _, params, prompts = _tiny(dev)
        ref = FlatParams(TINY, (params.flat.float() * (1 + 0.03 * torch.randn(params.flat.shape, generator=torch.Generator().manual_seed(3)).to(dev))).to(torch.bfloat16), params.specs)
        ge = GRPOEngine(TINY, params, GRPOHyper(num_generations=5, learning_rate=1e-4, trim_completions=trim), ref=ref)
        comps = [_completions(TINY, 5, C, [1, 7, C // 2, C, None], 11, dev), _completions(TINY, 5, C, [3, None, 2, 1, C - 1], 12, dev)]
        adv, _ = group_advantages(torch.tensor([2.0, 0.0, 1.0, 0.5, 1.5]), 5)
        with K.plan(gemm_no_split=1, gemm_tile=256):
            out = ge.score_and_backward_multi(prompts, comps, [adv, adv], grad_scale=0.5)
        ge.optimizer_step()
        m = out["mask"].bool()
        res[trim] = (out["logps"][m].clone(), out["ref_logps"][m].clone(), float(out["loss"]), float(out["kl"]), ge.policy.flat.clone(),
                     out["scored_tokens"], out["mask"].cpu(), out["logps"].cpu(), out["ref_logps"].cpu())
    assert res[True][5] == 1 + 7 + 8 + 16 + 16 + 3 + 16 + 2 + 1 + 15 and res[False][5] == 10 * C
    assert torch.equal(res[True][0], res[False][0]) and torch.equal(res[True][1], res[False][1])
    # (the loss kernel combines its per-row sums with fp32 atomics in arrival order: equal inputs, last-digit differences in the scalars)
    # (and the loss is a small difference of O(|A|) row terms -- the advantages of a group sum to zero -- so its tolerance is absolute)
    assert abs(res[True][2] - res[False][2]) <= 1e-6 and abs(res[True][3] - res[False][3]) <= 2e-6 * abs(res[False][3])
    # updated bf16 weights: equal except where the gradients' summation-order noise (see _check_pass) moved an fp32 master weight across
    # a bf16 rounding boundary -- at most one bf16 ulp (2^-7 relative), on a handful of the 2.9 M parameters
    wa, wb = res[True][4].float(), res[False][4].float()
    assert bool(((wa - wb).abs() <= 2.0 ** -7 * wb.abs() + 1e-30).all()) and float((wa != wb).float().mean()) < 1e-3
    # oracle restatement of TR:493-498, 551-552, 640-643 on the engine's own log-probs: the trimmed step's loss / KL
    mask_o = GR.completion_mask(torch.cat(comps, 0).cpu(), TINY.eos_token_id)
    assert torch.equal(mask_o.int(), res[True][6].int())
    adv2 = torch.cat([adv, adv])
    loss_o = GR.grpo_loss(res[True][7], res[True][8], adv2, mask_o, 0.04)
    assert abs(float(loss_o) - res[True][2]) <= 1e-5


def test_synthetic_lengths_rollout(dev):
    """SamplingParams.synthetic_lengths: EOS exactly at the scheduled token of every row, pad behind it, nowhere else; the graph-replayed
    loop equals the eager one; the loop stops early once every row finished."""
    _, params, prompts = _tiny(dev)
    eng = Qwen2VLEngine(TINY, params)
    roll = RolloutEngine(eng)
    C, Kn = 128, 4
    sp = SamplingParams(max_new_tokens=C, seed=7, synthetic_lengths=(1, 40))
    gen = torch.Generator().manual_seed(1_000_003 * sp.seed + 17)
    want = torch.randint(1, 41, (2 * Kn,), generator=gen)
    outs = []
    for use_graph in (False, True):
        st = {}
        with K.plan(skinny_blocks=1):
            out = roll.generate(prompts, Kn, sp, use_graph=use_graph, stats=st)
        mask, lengths = K.completion_mask(out, TINY.eos_token_id)
        assert lengths.cpu().tolist() == want.tolist()
        for b, n in enumerate(want.tolist()):
            assert int(out[b, n - 1]) == TINY.eos_token_id and bool((out[b, :n - 1] != TINY.eos_token_id).all())
            assert bool((out[b, n:] == TINY.pad_token_id).all())
        assert st["decode_steps"] <= 40 + 8 < C - 1                     # the loop left once all rows had finished (checked every 8 steps)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    with pytest.raises(ValueError):
        roll.generate(prompts, Kn, dataclasses.replace(sp, suppress_eos=True))
    # a length beyond max_new_tokens: that row never ends
    sp2 = SamplingParams(max_new_tokens=8, seed=7, synthetic_lengths=(9, 12))
    out = roll.generate(prompts, 2, sp2)
    assert bool((out != TINY.eos_token_id).all())
