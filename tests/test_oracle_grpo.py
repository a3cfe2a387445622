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
