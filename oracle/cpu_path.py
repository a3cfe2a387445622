"""CPU restatement of ONE FULL SG-RLVR / GRPO step for one prompt group, assembled from the oracle pieces
(oracle/qwen2vl_fp32.py model arithmetic + oracle/grpo_ref.py loss arithmetic).  TEST INFRASTRUCTURE ONLY: used by
tests/ (BASELINE.json configs[0], the "CPU eager plumbing" case), by scripts/run_cfg1_cpu.py and by bench.py's
``cpu_baseline`` leg (kind "port"), never by spacer_amd/.

It follows SGRLVRTrainer.compute_loss (SG_RLVR_trainer.py:384-686) phase by phase:
  rollout       :462-466  model.generate -> K sampled completions   (here: ViT + prefill ONCE, KV cache, top-k/top-p draw)
  mask          :493-498
  policy logps  :526-528  with grad          } scored as ONE shared-prompt packed sequence (the engine's algorithm:
  ref logps     :534-541  inference mode     } SURVEY 3.2 "work the native engine may legally skip"), not K padded rows
  KL, advantage, loss  :551-552, :632-643
  backward      HF Trainer.training_step -> loss.backward()  (autograd here)
so the CPU baseline is timed on the SAME algorithmic work the GPU path does (the reference's literal structure -- ViT x 3K,
prefill x K -- would be ~3x slower on the CPU; this port is the favourable one for the CPU).
"""
from __future__ import annotations

import math
import time
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

from . import grpo_ref as GR
from . import qwen2vl_fp32 as O


def _rope(x, cos, sin):
    return x * cos + O._rot_half(x) * sin


def prefill(w, cfg, embeds, pos3):
    """Causal forward of the prompt; returns (last-position logits, per-layer post-rotary K, V)."""
    col: Dict[str, List[torch.Tensor]] = {}
    hid = O.llm_forward(w, cfg, embeds, pos3, return_hidden=True, collect=col)
    logits = hid[-1:] @ O.lm_head_weight(w, cfg).float().t()
    return logits[0], col["k"], col["v"]


def decode_step(w, cfg, tok: torch.Tensor, pos: int, pk, pv, tk, tv, t: int):
    """One token for K sequences sharing the prompt K/V (pk[l]: (P, KV, hd)) with private tails tk[l]: (K, C, KV, hd)
    filled up to t.  tok (K,) -> logits (K, V); appends this step's k/v at tail position t."""
    H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    Kn, rep = tok.numel(), cfg["heads"] // cfg["kv_heads"]
    pos3 = torch.full((3, 1), pos, dtype=torch.long)
    cos, sin = O.mrope_tables(pos3, cfg)                       # (1, hd): the same position for every rollout
    x = w["model.embed_tokens.weight"].float()[tok]
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        h = O.rms_norm(x, w[p + "input_layernorm.weight"], cfg["rms_eps"])
        q = (h @ w[p + "self_attn.q_proj.weight"].float().t() + w[p + "self_attn.q_proj.bias"].float()).view(Kn, H, hd)
        k = (h @ w[p + "self_attn.k_proj.weight"].float().t() + w[p + "self_attn.k_proj.bias"].float()).view(Kn, KV, hd)
        v = (h @ w[p + "self_attn.v_proj.weight"].float().t() + w[p + "self_attn.v_proj.bias"].float()).view(Kn, KV, hd)
        q, k = _rope(q, cos[:, None, :], sin[:, None, :]), _rope(k, cos[:, None, :], sin[:, None, :])
        tk[i][:, t], tv[i][:, t] = k, v
        qg = q.view(Kn, KV, rep, hd)
        s_p = torch.einsum("kgrd,pgd->kgrp", qg, pk[i])                       # prompt keys, shared
        s_t = torch.einsum("kgrd,ktgd->kgrt", qg, tk[i][:, :t + 1])           # own tail
        a = torch.softmax(torch.cat([s_p, s_t], -1) / math.sqrt(hd), -1)
        P = pk[i].shape[0]
        o = torch.einsum("kgrp,pgd->kgrd", a[..., :P], pv[i]) + torch.einsum("kgrt,ktgd->kgrd", a[..., P:], tv[i][:, :t + 1])
        x = x + o.reshape(Kn, H * hd) @ w[p + "self_attn.o_proj.weight"].float().t()
        h = O.rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg["rms_eps"])
        g = F.silu(h @ w[p + "mlp.gate_proj.weight"].float().t()) * (h @ w[p + "mlp.up_proj.weight"].float().t())
        x = x + g @ w[p + "mlp.down_proj.weight"].float().t()
    x = O.rms_norm(x, w["model.norm.weight"], cfg["rms_eps"])
    return x @ O.lm_head_weight(w, cfg).float().t()


def sample(logits: torch.Tensor, gen: torch.Generator, top_k: int = 50, top_p: float = 0.95, greedy: bool = False):
    """temperature 1 -> top-k -> top-p -> multinomial (HF warper order, SG_RLVR_trainer.py:277-283 + era default top_k=50)."""
    if greedy:
        return logits.argmax(-1)
    v, idx = torch.topk(logits, min(top_k, logits.shape[-1]), dim=-1)
    pr = torch.softmax(v, -1)
    keep = (pr.cumsum(-1) - pr) < top_p                    # descending order: keep the smallest prefix reaching top_p
    pr = torch.where(keep, pr, torch.zeros_like(pr))
    pick = torch.multinomial(pr / pr.sum(-1, keepdim=True), 1, generator=gen)
    return idx.gather(-1, pick).squeeze(-1)


def group_mask(P: int, Kn: int, C: int) -> torch.Tensor:
    T = P + Kn * C
    m = torch.zeros(T, T, dtype=torch.bool)
    m[:P, :P] = torch.ones(P, P, dtype=torch.bool).tril()
    for k in range(Kn):
        a = P + k * C
        m[a:a + C, :P] = True
        m[a:a + C, a:a + C] = torch.ones(C, C, dtype=torch.bool).tril()
    return m


def group_logps(w, cfg, prompt_ids, comps, pixel_rows, grids):
    """(K, C) per-token log-probs with the prompt computed once (same numbers as O.completion_logps: test_oracle_model.py)."""
    P, (Kn, C) = prompt_ids.numel(), comps.shape
    ve = O.vit_forward(w, cfg, pixel_rows, grids) if pixel_rows is not None else None
    ids = torch.cat([prompt_ids, comps.reshape(-1)])
    e = O.embed_with_video(w, cfg, ids, ve)
    pos3, delta = O.mrope_position_ids(prompt_ids.tolist(), grids or [], cfg)
    cp = (P + delta + torch.arange(C)).view(1, C).expand(3, C)
    pos = torch.cat([pos3] + [cp] * Kn, 1)
    hid = O.llm_forward(w, cfg, e, pos, group_mask(P, Kn, C), return_hidden=True)
    sel = torch.stack([torch.where(torch.arange(C) == 0, torch.full((C,), P - 1), P + k * C + torch.arange(C) - 1) for k in range(Kn)])
    lg = hid[sel.reshape(-1)] @ O.lm_head_weight(w, cfg).float().t()
    return torch.log_softmax(lg, -1).gather(1, comps.reshape(-1, 1)).view(Kn, C)


def grpo_group_step(w, w_ref, cfg, prompt_ids, pixel_rows, grids, *, num_generations: int, max_new_tokens: int,
                    rewards: Optional[torch.Tensor] = None, beta: float = 0.04, eos_token_id: int = 151645, seed: int = 0,
                    suppress_eos: bool = True, grad_names: Optional[List[str]] = None) -> dict:
    """One prompt group through the whole step on the CPU.  ``w`` must hold leaf tensors with requires_grad for the names in
    ``grad_names`` (default: every floating tensor).  Returns completions, log-probs, loss and per-phase seconds."""
    Kn, C, P = num_generations, max_new_tokens, prompt_ids.numel()
    gen = torch.Generator().manual_seed(seed)
    times = {}
    t0 = time.perf_counter()
    with torch.no_grad():
        ve = O.vit_forward(w, cfg, pixel_rows, grids) if pixel_rows is not None else None
        e = O.embed_with_video(w, cfg, prompt_ids, ve)
        pos3, delta = O.mrope_position_ids(prompt_ids.tolist(), grids or [], cfg)
        first, pk, pv = prefill(w, cfg, e, pos3)
        times["vit+prefill"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        L, KV, hd = cfg["layers"], cfg["kv_heads"], cfg["head_dim"]
        tk = [torch.zeros(Kn, C, KV, hd) for _ in range(L)]
        tv = [torch.zeros(Kn, C, KV, hd) for _ in range(L)]
        comps = torch.zeros(Kn, C, dtype=torch.long)
        lg = first.view(1, -1).expand(Kn, -1).clone()
        for t in range(C):
            if suppress_eos:
                lg[:, eos_token_id] = float("-inf")
            tok = sample(lg, gen)
            comps[:, t] = tok
            if t + 1 < C:
                lg = decode_step(w, cfg, tok, P + delta + t, pk, pv, tk, tv, t)
        times["decode"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        mask = GR.completion_mask(comps, eos_token_id)
        ref_lp = group_logps(w_ref, cfg, prompt_ids, comps, pixel_rows, grids)
        times["ref scoring"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    lp = group_logps(w, cfg, prompt_ids, comps, pixel_rows, grids)
    times["policy scoring"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if rewards is None:
        rewards = torch.rand(Kn, generator=gen) * 2.0
    adv, _ = GR.group_advantages(rewards, Kn)
    loss = GR.grpo_loss(lp, ref_lp, adv, mask, beta)
    loss.backward()
    times["loss+backward"] = time.perf_counter() - t0
    return dict(completions=comps, logps=lp.detach(), ref_logps=ref_lp, loss=float(loss.detach()), mask=mask, seconds=times,
                total_seconds=sum(times.values()))
