"""CPU: hand-computed cases for the restated SG_RLVR_trainer.py arithmetic (oracle/grpo_ref.py)."""
import math

import torch

from oracle import grpo_ref as GR


def test_completion_mask_first_eos_inclusive():
    ids = torch.tensor([[5, 7, 9, 7], [5, 5, 5, 5], [7, 1, 1, 1]])
    assert GR.completion_mask(ids, 7).tolist() == [[1, 1, 0, 0], [1, 1, 1, 1], [1, 0, 0, 0]]


def test_k3_kl_values_and_clamp():
    ref, lp = torch.tensor([[0.0, -1.0, 20.0, -20.0]]), torch.zeros(1, 4)
    kl = GR.k3_kl(ref, lp)[0]
    assert kl[0] == 0
    assert abs(float(kl[1]) - (math.exp(-1) + 1 - 1)) < 1e-6
    assert abs(float(kl[2]) - (math.exp(10) - 10 - 1)) < 1e-2 and abs(float(kl[3]) - (math.exp(-10) + 10 - 1)) < 1e-5


def test_temporal_and_length_bonus():
    rpf = torch.tensor([[1.0, 1.0], [0.0, 1.0], [0.05, 0.0], [1.9, 1.0]])
    sh = torch.tensor([[1.0, 1.0], [1.0, 0.0]])
    r, t = GR.temporal_bonus(rpf, sh, True, True)          # mean acc .7375 < .8 * 1 -> no bonus
    assert t == 0.0 and torch.allclose(r, rpf.sum(1))
    sh2 = torch.tensor([[0.5, 1.0], [1.0, 0.0]])
    r, t = GR.temporal_bonus(rpf, sh2, True, True)         # .7375 >= .6 -> +0.3 where acc > .1
    assert t == 1.0 and torch.allclose(r, torch.tensor([2.3, 1.0, 0.05, 3.2]))
    r, t = GR.temporal_bonus(rpf, None, False, True)
    assert t == 0.5 and torch.allclose(r, rpf.sum(1))
    mask = torch.zeros(4, 600, dtype=torch.int32)
    for i, n in enumerate((320, 512, 400, 513)):
        mask[i, :n] = 1
    out = GR.length_bonus(rpf.sum(1), rpf, mask, True)     # rows 0,3 have acc>.1 (two of them) ; lens 320 ok, 513 not
    assert torch.allclose(out, torch.tensor([2.2, 1.0, 0.05, 2.9]))
    one = torch.tensor([[1.0, 0.0], [0.0, 0.0]])
    assert torch.allclose(GR.length_bonus(one.sum(1), one, mask[:2], True), one.sum(1))   # only ONE correct: no bonus


def test_group_advantages_unbiased_std_and_zero_std():
    r = torch.tensor([1.0, 2.0, 3.0, 4.0, 2.0, 2.0, 2.0, 2.0])
    adv, std = GR.group_advantages(r, 4)
    s = math.sqrt(((1.5 ** 2) * 2 + (0.5 ** 2) * 2) / 3)
    assert torch.allclose(std[:4], torch.full((4,), s)) and torch.allclose(adv[:4], (r[:4] - 2.5) / (s + 1e-4))
    assert torch.equal(adv[4:], torch.zeros(4)) and torch.equal(std[4:], torch.zeros(4))


def test_loss_and_gradient_closed_form():
    lp = torch.tensor([[-1.0, -2.0, -3.0], [-0.5, -0.5, -0.5]])
    ref = torch.tensor([[-1.5, -2.0, -30.0], [-0.5, -1.0, -0.5]])
    adv = torch.tensor([1.0, -2.0]); mask = torch.tensor([[1, 1, 0], [1, 1, 1]]); beta = 0.04
    loss, grad = GR.grpo_loss_and_grad(lp, ref, adv, mask, beta)
    kl = lambda x: math.exp(x) - x - 1  # noqa: E731
    row0 = (-(1 - beta * kl(-0.5)) - (1 - beta * kl(0.0))) / 2
    row1 = (-(-2 - beta * kl(0.0)) - (-2 - beta * kl(-0.5)) - (-2 - beta * kl(0.0))) / 3
    assert abs(float(loss) - (row0 + row1) / 2) < 1e-6
    # d/dlp = (-A + beta (1 - e^x)) * mask / len / G
    assert abs(float(grad[0, 0]) - (-1 + beta * (1 - math.exp(-0.5))) / 2 / 2) < 1e-7
    assert float(grad[0, 2]) == 0.0
    assert abs(float(grad[1, 1]) - (2 + beta * (1 - math.exp(-0.5))) / 3 / 2) < 1e-7


# ----------------------------------------------------------------------------------------------- the reference's own lines
# tests/golden/grpo_lines.json = inputs + outputs of SG_RLVR_trainer.py:493-498, 551-552, 598-643 EXECUTED from the reference's text
# (scripts/make_golden_grpo.py: line ranges cut out with asserted anchors, exec'd against a stub ``self``): 60 mask cases, 169 step
# cases (every flag combination, std = 0 groups, clamp edges at +-10, 0.8x threshold ties, the 320 / 512 length edges, exactly one
# correct rollout, two groups per batch).  Checked here: the oracle's restatement AND the product's host functions (spacer_amd/grpo.py).
def _golden():
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "grpo_lines.json")) as f:
        return json.load(f)


def _prefix_mask(lengths, C):
    m = torch.zeros(len(lengths), C, dtype=torch.int32)
    for k, n in enumerate(lengths):
        m[k, :n] = 1
    return m


def test_golden_mask_lines_493_498():
    G = _golden()
    assert G["meta"]["n_mask"] == len(G["mask"]) >= 60
    for c in G["mask"]:
        ids = torch.tensor(c["completion_ids"])
        assert GR.completion_mask(ids, c["eos_token_id"]).tolist() == c["completion_mask"]


def test_golden_step_lines_551_643_oracle_and_host_functions():
    from spacer_amd import grpo as HOST
    G = _golden()
    assert len(G["step"]) >= 160
    seen = set()
    for c in G["step"]:
        Kn, C = c["num_generations"], c["C"]
        rpf = torch.tensor(c["rewards_per_func"], dtype=torch.float32)
        srpf = None if c["shuffled_rewards_per_func"] is None else torch.tensor(c["shuffled_rewards_per_func"], dtype=torch.float32)
        mask = _prefix_mask(c["completion_lengths"], C)
        lp, ref = torch.tensor(c["per_token_logps"], dtype=torch.float32), torch.tensor(c["ref_per_token_logps"], dtype=torch.float32)
        want_r, want_a = torch.tensor(c["rewards"], dtype=torch.float32), torch.tensor(c["advantages"], dtype=torch.float32)
        # ---- oracle restatement (bit-equal on the reward path: same fp32 torch ops in the same order)
        r_o, t_o = GR.temporal_bonus(rpf, srpf, c["temporal"], c["video"])
        r_o = GR.length_bonus(r_o, rpf, mask, c["len_control"])
        a_o, _ = GR.group_advantages(r_o, Kn)
        assert t_o == c["temporal_rewards"], c["tag"]
        assert torch.equal(r_o, want_r), (c["tag"], r_o, want_r)
        assert torch.equal(a_o, want_a), (c["tag"], a_o, want_a)
        if c["per_token_kl"] is not None:
            assert torch.equal(GR.k3_kl(ref, lp), torch.tensor(c["per_token_kl"], dtype=torch.float32)), c["tag"]
        loss_o, g_o = GR.grpo_loss_and_grad(lp, ref, want_a, mask, c["beta"])
        assert torch.equal(loss_o, torch.tensor(c["loss"], dtype=torch.float32)) or abs(float(loss_o) - c["loss"]) <= 1e-6 * abs(c["loss"]), c["tag"]
        assert torch.allclose(g_o, torch.tensor(c["dlogp"], dtype=torch.float32), rtol=1e-6, atol=1e-9), c["tag"]
        # ---- the product's host functions (spacer_amd/grpo.py: what SGRLVRTrainer calls)
        r_h, t_h = HOST.temporal_bonus(rpf, srpf, c["temporal"], c["video"])
        r_h = HOST.length_bonus(r_h, rpf, torch.tensor(c["completion_lengths"]), c["len_control"])
        a_h, _ = HOST.group_advantages(r_h, Kn)
        assert t_h == c["temporal_rewards"] and torch.equal(r_h, want_r) and torch.equal(a_h, want_a), c["tag"]
        seen.add((c["temporal"], c["video"], c["len_control"]))
    assert len(seen) == 8                                  # every flag combination is in the table


def test_golden_metric_lines_650_683_packed_gather():
    """The logged metrics (TR:650-683: nine ``gather_for_metrics`` calls per micro-batch) executed from the reference's text on emulated
    worlds of 1 / 2 / 3 / 8 ranks (one prompt group per rank) vs the product's ONE packed vector per rank (``pack_metrics``) and its
    reduction (``reduce_metrics``): same keys, same values (means of equal-sized rank means re-associate an fp32 sum: 1e-6 relative);
    ``all_wrong`` / ``all_correct`` are fractions of ranks, exact."""
    from spacer_amd.open_r1.trainer import SG_RLVR_trainer as T
    G = _golden()
    assert G["meta"]["n_metrics"] == len(G["metrics"]) >= 40
    worlds = set()
    for c in G["metrics"]:
        packed = []
        for r in c["ranks"]:
            mask = _prefix_mask(r["completion_lengths"], c["C"])
            kl = torch.tensor(r["per_token_kl"], dtype=torch.float32)
            mean_kl = float(((kl * mask).sum(1) / mask.sum(1)).mean())
            packed.append(T.pack_metrics(torch.tensor(r["completion_lengths"]), torch.tensor(r["rewards_per_func"], dtype=torch.float32),
                                         torch.tensor(r["rewards"], dtype=torch.float32), r["temporal_rewards"],
                                         torch.tensor(r["std_grouped_rewards"], dtype=torch.float32), mean_kl))
        got = T.reduce_metrics(torch.stack(packed), ["accuracy_reward", "format_reward"], c["temporal"])
        want = c["metrics"]
        assert list(got) == list(want), (list(got), list(want))             # same keys in the reference's order
        for k in want:
            if k in ("all_wrong", "all_correct"):
                assert got[k] == want[k], (k, got[k], want[k])
            else:
                assert abs(got[k] - want[k]) <= 2e-6 * max(abs(want[k]), 1e-3), (k, got[k], want[k], c["world"])
        worlds.add(c["world"])
    assert worlds == {1, 2, 3, 8}


def test_golden_logp_lines_357_366_and_the_completion_slice():
    """TR:357-366 executed on a stub model's logits (+ TR:528, the ``[:, prompt_length - 1:]`` slice): the oracle's ``per_token_logps``
    restatement, row by row."""
    from oracle import qwen2vl_fp32 as O
    G = _golden()
    assert G["meta"]["n_logps"] == len(G["logps"]) >= 24
    for c in G["logps"]:
        logits, ids = torch.tensor(c["logits"], dtype=torch.float32), torch.tensor(c["input_ids"])
        want = torch.tensor(c["per_token_logps"], dtype=torch.float32)
        got = torch.stack([O.per_token_logps(logits[b], ids[b]) for b in range(ids.shape[0])])
        assert torch.equal(got, want) or float((got - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max()))
        assert torch.equal(want[:, c["prompt_length"] - 1:], torch.tensor(c["completion_logps"], dtype=torch.float32))
