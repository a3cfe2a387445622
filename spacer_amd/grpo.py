"""SG-RLVR / GRPO step on the HIP engine: rollout -> reward shaping -> policy/ref scoring -> loss -> backward ->
data-parallel gradient reduction -> AdamW.  This is the arithmetic of ``SGRLVRTrainer.compute_loss``
(SG_RLVR_trainer.py:384-686) plus the optimizer step HF Trainer / DeepSpeed run after it, with the reference's
redundancy removed (ViT once per weights, prompt prefilled once, logits only on completion rows).

Host-side reward shaping (T-GRPO bonus :598-617, length bonus :620-629, group advantage :632-638) stays
plain Python/torch on K-element tensors, as in the reference; everything token-sized runs in libspacer_hip.so.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import kernels as K
from .qwen2vl.config import Qwen2VLConfig
from .qwen2vl.engine import Qwen2VLEngine
from .qwen2vl.weights import FlatParams
from .rollout import PromptInput, RolloutEngine, SamplingParams

F32 = torch.float32


@dataclass
class GRPOHyper:
    num_generations: int = 8          # --num_generations 8   (run_SpaceR_SG_RLVR.sh:39)
    beta: float = 0.04                # --beta 0.04           (:36)
    learning_rate: float = 1e-6       # --learning_rate 1e-6  (:22)
    weight_decay: float = 0.01        # --weight_decay 0.01   (:24)
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_eps: float = 1e-8
    max_grad_norm: float = 5.0        # --max_grad_norm 5     (:37)
    temporal: bool = True             # --temporal true       (:19)
    len_control: bool = True          # --len_control true    (:20)
    lr_scheduler_type: str = "cosine"  # (:23)
    total_steps: int = 1000
    warmup_steps: int = 0
    grad_comm_bf16: bool = False      # gradient all-reduce on a bf16 wire (DeepSpeed bf16's communication dtype); fp32 when False


# ------------------------------------------------------------------------------------- reward shaping (host)
def temporal_bonus(rewards_per_func: torch.Tensor, shuffled_rewards_per_func: Optional[torch.Tensor], temporal: bool,
                   has_video: bool) -> Tuple[torch.Tensor, float]:
    """SG_RLVR_trainer.py:598-617.  Column 0 = accuracy reward.  Returns (summed rewards [G], temporal_reward)."""
    if temporal and has_video:
        t = rewards_per_func.clone()
        if float(t[:, 0].mean()) >= 0.8 * float(shuffled_rewards_per_func[:, 0].mean()):
            sel = t[:, 0] > 0.1
            t[sel, 0] += 0.3
            return t.sum(dim=1), 1.0
        return t.sum(dim=1), 0.0
    return rewards_per_func.sum(dim=1), 0.5


def length_bonus(rewards: torch.Tensor, rewards_per_func: torch.Tensor, lengths: torch.Tensor, len_control: bool) -> torch.Tensor:
    """SG_RLVR_trainer.py:620-629: +0.2 for 320 <= len <= 512, only when more than one rollout has acc > 0.1."""
    out = rewards.clone()
    if len_control:
        idx = [i for i in range(rewards.numel()) if float(rewards_per_func[i, 0]) > 0.1]
        if len(idx) > 1:
            for i in idx:
                if 320 <= int(lengths[i]) <= 512:
                    out[i] += 0.2
    return out


def group_advantages(rewards: torch.Tensor, num_generations: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """SG_RLVR_trainer.py:632-638: (r - mean_g) / (std_g + 1e-4) with torch's unbiased std."""
    g = rewards.view(-1, num_generations)
    mean = g.mean(dim=1, keepdim=True)
    std = g.std(dim=1, keepdim=True)
    return ((g - mean) / (std + 1e-4)).reshape(-1), std.expand_as(g).reshape(-1)


def lr_at(step: int, h: GRPOHyper) -> float:
    """HF get_scheduler semantics for "cosine" / "linear" / "constant" with optional warm-up; step counts from 0."""
    if step < h.warmup_steps:
        return h.learning_rate * (step + 1) / max(1, h.warmup_steps)
    if h.lr_scheduler_type == "constant":
        return h.learning_rate
    prog = (step - h.warmup_steps) / max(1, h.total_steps - h.warmup_steps)
    prog = min(max(prog, 0.0), 1.0)
    if h.lr_scheduler_type == "linear":
        return h.learning_rate * (1.0 - prog)
    return h.learning_rate * 0.5 * (1.0 + math.cos(math.pi * prog))


def allreduce_flat_(flat: torch.Tensor, pg, bucket_elems: int = 1 << 28, wire_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """SUM all-reduce of one flat tensor in a few large buckets (1 GiB of fp32 each by default), all in flight at
    once.  On MI355X the 8 GPUs are fully connected by point-to-point xGMI links, so collectives are per-link bound:
    few, big messages; the mean (1/world) is folded into the optimizer's grad_scale instead of a second pass.

    wire_dtype = torch.bfloat16 halves the bytes on the links: each bucket is rounded to bf16, summed in bf16 by the
    collective and widened back into ``flat`` -- what the reference's DeepSpeed bf16 configuration does with its gradient
    buckets (communication in the training dtype).  Default: exchange the fp32 values themselves."""
    import torch.distributed as dist
    if wire_dtype is None or wire_dtype == flat.dtype:
        works = [dist.all_reduce(flat[a:a + bucket_elems], group=pg, async_op=True) for a in range(0, flat.numel(), bucket_elems)]
        for w in works:
            w.wait()
        return flat
    pending = []
    for a in range(0, flat.numel(), bucket_elems):
        wire = flat[a:a + bucket_elems].to(wire_dtype)
        pending.append((a, wire, dist.all_reduce(wire, group=pg, async_op=True)))
    for a, wire, w in pending:
        w.wait()
        flat[a:a + wire.numel()].copy_(wire)
    return flat


# ------------------------------------------------------------------------------------- the step engine
class GRPOEngine:
    """Owns policy + frozen reference weights, fp32 master / Adam state / gradients, and runs the step phases."""

    def __init__(self, cfg: Qwen2VLConfig, policy: FlatParams, hyper: GRPOHyper, *, ref: Optional[FlatParams] = None,
                 process_group=None, cache_wT: bool = True):
        self.cfg, self.h = cfg, hyper
        self.dev = policy.flat.device
        self.policy = policy
        self.ref = ref if ref is not None else FlatParams(cfg, policy.flat.clone(), policy.specs)   # create_reference_model
        # W^T copies for the dX GEMMs are kept for the whole optimizer step (all prompt groups reuse them): +1x weights
        self.engine = Qwen2VLEngine(cfg, self.policy, cache_wT=cache_wT)
        self.ref_engine = Qwen2VLEngine(cfg, self.ref)
        self.roll = RolloutEngine(self.engine)
        self.master = FlatParams(cfg, policy.flat.float(), policy.specs)
        self.G = policy.like(F32)
        self.m = torch.zeros_like(self.master.flat)
        self.v = torch.zeros_like(self.master.flat)
        self.step_count = 0
        self.pg = process_group
        self._sumsq = torch.zeros(1, device=self.dev, dtype=F32)

    # -------------------------------------------------------------- phases
    def rollout(self, prompts: List[PromptInput], sp: SamplingParams, stats: Optional[dict] = None) -> torch.Tensor:
        return self.roll.generate(prompts, self.h.num_generations, sp, stats=stats)

    def score_and_backward(self, prompt: PromptInput, completion_ids: torch.Tensor, advantages: torch.Tensor,
                           grad_scale: float = 1.0, *, era_rule: bool = False) -> Dict[str, torch.Tensor]:
        """One prompt group: masks, reference + policy log-probs, loss, backward into self.G.  ``advantages`` fp32 [K]
        (device).  grad_scale folds 1/num_groups (gradient accumulation mean)."""
        cfg = self.cfg
        mask, lengths = K.completion_mask(completion_ids, cfg.eos_token_id)
        with torch.no_grad():
            ref_lp = self.ref_engine.score_group(prompt.ids, completion_ids, prompt.pix, prompt.grids, era_rule=era_rule)
            tape: dict = {}
            lp = self.engine.score_group(prompt.ids, completion_ids, prompt.pix, prompt.grids, tape=tape, era_rule=era_rule)
            loss, kl, dlogp = K.grpo_loss(lp, ref_lp, advantages, mask, self.h.beta)
            if grad_scale != 1.0:
                dlogp.mul_(grad_scale)
            self.engine.backward_group(tape, dlogp, self.G)
        return dict(loss=loss, kl=kl, logps=lp, ref_logps=ref_lp, mask=mask, lengths=lengths)

    def sft_forward_backward(self, ids: torch.Tensor, pix, grids, label_mask: torch.Tensor, *, grad_scale: float = 1.0,
                             second_per_grid_ts=None) -> float:
        """Supervised objective of open_r1/sft.py on the same kernels: mean cross-entropy of ids[1:] over the positions whose
        label is kept (``label_mask`` bool/0-1 [S], True where labels != -100: the reference masks pad and visual tokens,
        sft.py:170-181), gradients accumulated into G scaled by grad_scale.  Returns the loss."""
        tape = {}
        logp = self.engine.score_sequence(ids, pix, grids, tape=tape, second_per_grid_ts=second_per_grid_ts)
        m = label_mask.reshape(-1)[1:].to(self.dev, torch.float32)
        n = float(m.sum().clamp_min(1.0))
        loss = float(-(logp * m).sum() / n)
        self.engine.backward_group(tape, (-(m / n) * grad_scale).view(1, -1), self.G)
        return loss

    def reduce_gradients(self) -> None:
        """Data-parallel exchange: SUM all-reduce of the flat fp32 gradient over RCCL in large buckets (xGMI is
        per-link bound: few, big collectives).  The mean is folded into the optimizer's grad_scale."""
        if self.pg is not None:
            allreduce_flat_(self.G.flat, self.pg, wire_dtype=torch.bfloat16 if self.h.grad_comm_bf16 else None)

    def optimizer_step(self, world_size: int = 1) -> float:
        """Global-norm clip (max_grad_norm) + AdamW on fp32 master, bf16 policy refreshed in the same kernel."""
        h = self.h
        lr = lr_at(self.step_count, h)
        self.step_count += 1
        gscale = 1.0 / world_size
        self._sumsq.zero_()
        K.sumsq_(self.G.flat, self._sumsq)
        K.adamw_step_(self.master.flat, self.policy.flat, self.m, self.v, self.G.flat, lr=lr, beta1=h.adam_beta1,
                      beta2=h.adam_beta2, eps=h.adam_eps, weight_decay=h.weight_decay, step=self.step_count,
                      sumsq=self._sumsq, max_norm=h.max_grad_norm, grad_scale=gscale)
        self.G.flat.zero_()
        self.engine.invalidate_cache()
        self.roll.invalidate()
        return lr

    def grad_norm(self, world_size: int = 1) -> float:
        return float(self._sumsq.sqrt()) / world_size
