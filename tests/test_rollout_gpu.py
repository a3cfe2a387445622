"""GPU: rollout (prefill + batched decode + sampler, eager and hipGraph) and one full GRPO step on the tiny model.

Sampling cannot be bit-compared with HF's RNG stream; parity is defined on teacher-forced quantities:
  * greedy decode (top_k = 1): every emitted token must be an arg-max of the ORACLE's next-token logits for the
    emitted prefix, up to the bf16 noise budget (oracle logit of the chosen token within 3e-2 of the oracle max);
  * the decode path (skinny GEMMs, KV cache, decode attention) and the scoring path (packed prefill kernels) must
    agree on the log-prob of the same tokens within 1e-2;
  * hipGraph replay == eager launch sequence, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny                      # noqa: E402
from oracle import grpo_ref as GR                      # noqa: E402
from oracle import qwen2vl_fp32 as O                   # noqa: E402
from spacer_amd import kernels as K                    # noqa: E402
from spacer_amd.grpo import GRPOEngine, GRPOHyper, group_advantages, length_bonus, temporal_bonus  # noqa: E402
from spacer_amd.qwen2vl.config import TINY             # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict  # noqa: E402
from spacer_amd.rollout import PromptInput, RolloutEngine, SamplingParams  # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine    # noqa: E402


@pytest.fixture(scope="module")
def tiny(dev):
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    rows, _ = O.patchify_frames(g["frames"], g["cfg"])
    prompts = [PromptInput(g["prompt"].to(dev), pix, [tuple(grid)]), PromptInput(g["prompt"][-9:].to(dev), None, None)]
    return dict(g=g, params=params, wb=wb, prompts=prompts, rows=rows.to(torch.bfloat16).float(), grid=tuple(grid))


@pytest.mark.parametrize("Kn", [2, 10])       # 4 rows: the <= 16-row decode forms; 20 rows: the 17..64-row forms
def test_greedy_rollout_is_oracle_argmax(tiny, dev, Kn):
    eng = Qwen2VLEngine(TINY, tiny["params"])
    roll = RolloutEngine(eng)
    sp = SamplingParams(max_new_tokens=7, top_k=1, top_p=1.0, suppress_eos=True)
    out = roll.generate(tiny["prompts"], Kn, sp, use_graph=False)
    assert out.shape == (2 * Kn, 7)
    assert all(torch.equal(out[0], out[i]) and torch.equal(out[Kn], out[Kn + i]) for i in range(Kn))   # greedy: the K rollouts coincide
    g = tiny["g"]
    for pi, (pids, rows, grids) in enumerate(((g["prompt"], tiny["rows"], [tiny["grid"]]), (g["prompt"][-9:], None, None))):
        comp = out[Kn * pi].cpu()
        ids = torch.cat([pids, comp])
        lg = O.full_logits(tiny["wb"], g["cfg"], ids, rows, grids)
        P = pids.numel()
        for t in range(7):
            row = lg[P - 1 + t].clone()
            row[TINY.eos_token_id] = float("-inf")
            assert float(row.max() - row[comp[t]]) < 3e-2, (pi, t, float(row.max() - row[comp[t]]))


@pytest.mark.parametrize("Kn", [3, 12])
def test_graph_replay_equals_eager_and_scoring_agrees(tiny, dev, monkeypatch, Kn):
    # split-K flushes of the decode GEMMs are fp32 atomics (arrival order = last-bit noise in the residual stream, which
    # can flip a sampled token whose draw sits on a CDF boundary); one K range per block makes the sums reproducible so
    # that "replay == eager" can be asserted bit for bit
    monkeypatch.setattr(K.PLAN, "skinny_blocks", 1)
    eng = Qwen2VLEngine(TINY, tiny["params"])
    roll = RolloutEngine(eng)
    sp = SamplingParams(max_new_tokens=12, top_k=50, top_p=0.95, seed=11, suppress_eos=True)
    a = roll.generate(tiny["prompts"], Kn, sp, use_graph=False)
    st = {}
    b = roll.generate(tiny["prompts"], Kn, sp, use_graph=True, stats=st)
    assert st["graph"] and torch.equal(a, b)
    assert len({tuple(r.tolist()) for r in a[:3]}) > 1                           # sampling: rollouts of one prompt differ
    # log-prob of the sampled tokens: scoring path (packed, shared prefix) vs the oracle
    lp = eng.score_group(tiny["prompts"][0].ids, a[:3], tiny["prompts"][0].pix, tiny["prompts"][0].grids)
    want = O.completion_logps(tiny["wb"], tiny["g"]["cfg"], tiny["g"]["prompt"], a[:3].cpu(), tiny["rows"], [tiny["grid"]])
    assert (lp.cpu() - want).abs().max() < 1e-2


def test_eos_stops_and_pads(tiny, dev):
    eng = Qwen2VLEngine(TINY, tiny["params"])
    roll = RolloutEngine(eng)
    sp = SamplingParams(max_new_tokens=40, top_k=50, top_p=0.95, seed=3)
    out = roll.generate(tiny["prompts"][1:], 8, sp)
    mask, lens = K.completion_mask(out, TINY.eos_token_id)
    for r in range(out.shape[0]):
        n = int(lens[r])
        if n < out.shape[1]:
            assert int(out[r, n - 1]) == TINY.eos_token_id
            assert bool((out[r, n:] == TINY.pad_token_id).all())


def test_full_grpo_step_matches_oracle(tiny, dev):
    """loss, KL, d loss/d logp and the AdamW update direction of one step, against the restated trainer math."""
    g = tiny["g"]
    hyper = GRPOHyper(num_generations=3, beta=0.04, learning_rate=1e-3, max_grad_norm=5.0, temporal=False, len_control=True)
    params = FlatParams(TINY, tiny["params"].flat.clone(), tiny["params"].specs)
    ge = GRPOEngine(TINY, params, hyper)
    # perturb the policy so that policy != ref (KL term active)
    torch.manual_seed(0)
    ge.policy["llm.1.down_w"].add_(torch.randn_like(ge.policy["llm.1.down_w"]) * 0.02)
    ge.master["llm.1.down_w"].copy_(ge.policy["llm.1.down_w"].float())
    comp = g["completions"].clone()
    comp[1, 3] = TINY.eos_token_id                                               # one rollout stops early
    rpf = torch.tensor([[1.0, 1.0], [0.0, 1.0], [1.9, 0.0]])
    rewards, _ = temporal_bonus(rpf, None, hyper.temporal, True)
    mask_o = GR.completion_mask(comp, TINY.eos_token_id)
    rewards = length_bonus(rewards, rpf, mask_o.sum(1), hyper.len_control)
    adv, _ = group_advantages(rewards, 3)
    adv_o, _ = GR.group_advantages(GR.length_bonus(GR.temporal_bonus(rpf, None, False, True)[0], rpf, mask_o, True), 3)
    assert torch.allclose(adv, adv_o)
    pr = tiny["prompts"][0]
    before = ge.master["llm.1.down_w"].clone()
    res = ge.score_and_backward(pr, comp.to(dev), adv.to(dev))
    # oracle
    wpol = {k: v.float().cpu() for k, v in export_state_dict(ge.policy).items()}
    wpol["visual.patch_embed.proj.weight"] = wpol["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    lp_o = O.completion_logps(wpol, g["cfg"], g["prompt"], comp, tiny["rows"], [tiny["grid"]])
    ref_o = O.completion_logps(tiny["wb"], g["cfg"], g["prompt"], comp, tiny["rows"], [tiny["grid"]])
    loss_o, dlp_o = GR.grpo_loss_and_grad(lp_o, ref_o, adv, mask_o, hyper.beta)
    assert torch.equal(res["mask"].cpu(), mask_o)
    assert abs(float(res["loss"]) - float(loss_o)) < 5e-3
    assert abs(float(res["kl"]) - float(GR.kl_metric(lp_o, ref_o, mask_o))) < 5e-3
    gnorm_before = float(ge.G.flat.double().pow(2).sum().sqrt())
    assert gnorm_before > 0
    ge.optimizer_step()
    assert abs(ge.grad_norm() - gnorm_before) < 1e-3 * gnorm_before
    assert float(ge.G.flat.abs().max()) == 0.0
    delta = ge.master["llm.1.down_w"] - before
    assert float(delta.abs().max()) > 0 and float(delta.abs().max()) <= 1.01e-3 * 1.2   # |AdamW step 1| ~ lr
    assert torch.equal(ge.policy["llm.1.down_w"], ge.master["llm.1.down_w"].to(torch.bfloat16))


def test_more_than_64_rows_decode_as_one_batch(tiny, dev):
    """65..128 rows go through the two-block skinny GEMMs in ONE decode batch; greedy tokens equal the <= 64-row chunks'."""
    eng = Qwen2VLEngine(TINY, tiny["params"])
    roll = RolloutEngine(eng)
    sp = SamplingParams(max_new_tokens=6, top_k=1, top_p=1.0, suppress_eos=True)
    prompts = [tiny["prompts"][0], tiny["prompts"][1], tiny["prompts"][0]]
    st = {}
    big = roll.generate(prompts, 24, sp, use_graph=True, stats=st)                 # 72 rows, one batch
    assert big.shape == (72, 6) and st["decode_steps"] == 5
    roll.MAX_ROWS = 64                                                           # instance override: chunks of 2 + 1 prompts
    small = roll.generate(prompts, 24, sp, use_graph=False)
    assert torch.equal(big, small)
    assert torch.equal(big[:24], big[48:])                                      # same prompt, greedy -> same rollouts


def test_per_prompt_generation_counts_share_the_prefill(tiny, dev):
    """Round 6: ``generate(prompts, [k_0, k_1, ...])`` -- the T-GRPO twins take G // 2 rollouts (TR:473), so a step's main + twin rollouts
    decode as fewer rows: prompt p owns the rows [row0[p], row0[p + 1]) of the batch (``spacer_attn_decode_shared_rows``).  Greedy
    decoding makes every rollout of a prompt THE arg-max continuation, which must not depend on how many siblings it has or on where its
    rows sit: rows of a [4, 2] / [3, 6] batch == the rows of uniform batches of the same prompts; eager == graph; the prefill runs once
    per prompt (its kept tape has one slice per prompt)."""
    eng = Qwen2VLEngine(TINY, tiny["params"])
    roll = RolloutEngine(eng)
    sp = SamplingParams(max_new_tokens=9, top_k=1, top_p=1.0, suppress_eos=True)
    with K.plan(skinny_blocks=1):
        ref = roll.generate(tiny["prompts"], 2, sp, use_graph=False)                    # [2 prompts x 2, 9]
        outs = [roll.generate(tiny["prompts"], [4, 2], sp, use_graph=g) for g in (False, True)]
        mixed3 = roll.generate(tiny["prompts"], [3, 6], sp, use_graph=False)            # 3 x 2 = 6 and 6 x 2 = 12 attention columns
    assert outs[0].shape == (6, 9) and torch.equal(outs[0], outs[1])
    for r in range(4):
        assert torch.equal(outs[0][r], ref[0])                                         # prompt 0: four copies of its greedy continuation
    for r in (4, 5):
        assert torch.equal(outs[0][r], ref[2])                                         # prompt 1 (text-only): two copies
    assert mixed3.shape == (9, 9) and all(torch.equal(mixed3[r], ref[0]) for r in range(3)) and all(torch.equal(mixed3[r], ref[2]) for r in range(3, 9))
    # sampled rollouts: shapes, pads behind EOS, and the prefill tape kept per real prompt
    roll.keep_prefill_tape = True
    vid = [tiny["prompts"][0], PromptInput(tiny["prompts"][0].ids, tiny["prompts"][0].pix, tiny["prompts"][0].grids)]
    out = roll.generate(vid, [4, 2], SamplingParams(max_new_tokens=12, seed=5))
    assert out.shape == (6, 12) and vid[0].prefill is not None and vid[1].prefill.index == 1 and vid[0].prefill.shared is vid[1].prefill.shared
    with pytest.raises(ValueError):
        roll.generate(tiny["prompts"][:1], 200, sp)
