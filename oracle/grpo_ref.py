"""CPU oracle: the reward-shaping / advantage / loss arithmetic of SGRLVRTrainer.compute_loss.

TEST INFRASTRUCTURE ONLY (see oracle/qwen2vl_fp32.py header for the import rule).

Each function restates, in plain torch fp32 on CPU, one block of
/root/reference/SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py (abbrev. TR):

  completion_mask      TR:493-498   first-EOS mask (inclusive of the EOS token)
  k3_kl                TR:551-552   exp(x) - x - 1, x = clamp(ref - pol, -10, 10)
  temporal_bonus       TR:598-617   T-GRPO: +0.3 on acc > 0.1 when mean(acc) >= 0.8 mean(shuffled acc)
  length_bonus         TR:620-629   +0.2 when 320 <= len <= 512, only if >1 rollouts have acc > 0.1
  group_advantages     TR:632-638   (r - mean_g) / (std_g + 1e-4), unbiased std
  grpo_loss            TR:640-643   -(exp(lp - sg(lp)) * A - beta * kl), masked per-row mean, mean over rows
  kl_metric            TR:682       masked mean KL

Pinning: the reference has no tests for these lines; they are ~40 lines of elementwise torch,
restated line by line and checked with hand-computed cases in tests/test_oracle_grpo.py
(autograd of this restatement is the oracle for the analytic gradient the HIP kernel emits).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch


def completion_mask(completion_ids: torch.Tensor, eos_token_id: int) -> torch.Tensor:
    is_eos = completion_ids == eos_token_id
    eos_idx = torch.full((is_eos.size(0),), is_eos.size(1), dtype=torch.long)
    has = is_eos.any(dim=1)
    eos_idx[has] = is_eos.int().argmax(dim=1)[has]
    seq = torch.arange(is_eos.size(1)).expand(is_eos.size(0), -1)
    return (seq <= eos_idx.unsqueeze(1)).int()


def k3_kl(ref_logps: torch.Tensor, logps: torch.Tensor) -> torch.Tensor:
    x = torch.clamp(ref_logps - logps, min=-10, max=10)
    return torch.exp(x) - x - 1


def temporal_bonus(rewards_per_func: torch.Tensor, shuffled_rewards_per_func: Optional[torch.Tensor],
                   temporal: bool, has_video: bool) -> Tuple[torch.Tensor, float]:
    """Returns (rewards (G,), temporal_reward scalar).  Column 0 is the accuracy reward."""
    if temporal and has_video:
        t = rewards_per_func.clone()
        if t[:, 0].mean() >= 0.8 * shuffled_rewards_per_func[:, 0].mean():
            m = t[:, 0] > 0.1
            t[m, 0] = t[m, 0] + 0.3
            tr = 1.0
        else:
            tr = 0.0
        return t.sum(dim=1), tr
    return rewards_per_func.sum(dim=1), 0.5


def length_bonus(rewards: torch.Tensor, rewards_per_func: torch.Tensor, mask: torch.Tensor,
                 len_control: bool) -> torch.Tensor:
    rewards = rewards.clone()
    if len_control:
        sel = torch.nonzero(rewards_per_func[:, 0] > 0.1, as_tuple=True)[0].tolist()
        lens = mask.sum(1)
        if len(sel) > 1:
            for i in sel:
                if 320 <= int(lens[i]) <= 512:
                    rewards[i] += 0.2
    return rewards


def group_advantages(rewards: torch.Tensor, num_generations: int) -> Tuple[torch.Tensor, torch.Tensor]:
    g = rewards.view(-1, num_generations)
    mean = g.mean(dim=1).repeat_interleave(num_generations, dim=0)
    std = g.std(dim=1).repeat_interleave(num_generations, dim=0)
    return (rewards - mean) / (std + 1e-4), std


def grpo_loss(logps: torch.Tensor, ref_logps: torch.Tensor, advantages: torch.Tensor,
              mask: torch.Tensor, beta: float) -> torch.Tensor:
    kl = k3_kl(ref_logps, logps)
    ptl = torch.exp(logps - logps.detach()) * advantages.unsqueeze(1)
    ptl = -(ptl - beta * kl)
    return ((ptl * mask).sum(dim=1) / mask.sum(dim=1)).mean()


def kl_metric(logps, ref_logps, mask) -> torch.Tensor:
    kl = k3_kl(ref_logps, logps)
    return ((kl * mask).sum(dim=1) / mask.sum(dim=1)).mean()


def grpo_loss_and_grad(logps, ref_logps, advantages, mask, beta):
    """Loss and d loss / d logps via autograd of the restatement above."""
    lp = logps.detach().clone().float().requires_grad_(True)
    loss = grpo_loss(lp, ref_logps.float(), advantages.float(), mask, beta)
    loss.backward()
    return loss.detach(), lp.grad.detach()
