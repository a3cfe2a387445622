"""CPU emulation of THIS ENGINE'S numerics: the arithmetic of oracle/qwen2vl_fp32.py with a bf16 rounding inserted at
exactly the places where spacer_amd/qwen2vl/engine.py stores a bf16 tensor (everything else -- accumulators, the residual
stream, norm statistics, softmax statistics, logits, log-sum-exp -- is fp32 on the GPU and fp32 here).
TEST INFRASTRUCTURE ONLY (used by tests/ and scripts/logp_error_budget.py; never by spacer_amd/).

Purpose: the per-operator error budget of the per-token log-probs (SG_RLVR_trainer.py:353-366, the quantity the
north-star pins at 1e-3).  Every rounding point belongs to a class; ``points`` selects which classes round, so the table
"error with only class X rounding" / "error with every class but X rounding" can be produced on the CPU at any depth.

Rounding classes (the bf16 tensors of the engine):
  norm    output of every LayerNorm / RMSNorm that feeds a GEMM (h, h2, merger ln, final norm of the LLM = ``final``)
  gemm    bf16 GEMM outputs: qkv (+bias), fc1 / gate|up, merger m1
  rope    q/k after the in-place rotary (computed in fp32 from the bf16 qkv, rounded back)
  p       softmax probabilities as the PV MFMA operand (un-normalised exp(s - m), the flash form)
  o       attention output o
  act     activation outputs (quick_gelu(f1), silu(g)*u, gelu(m1))
  vit_out merged video embeddings handed to the LLM
  final   the final-norm rows gathered for lm_head (hsel)
A class listed in ``split`` is kept as a hi+lo bf16 PAIR instead (value rounded to 16 mantissa bits): what a two-pass
MFMA on that operand computes.
"""
from __future__ import annotations

import math
from typing import Iterable, Optional

import torch
import torch.nn.functional as F

from . import qwen2vl_fp32 as O

ALL_POINTS = ("norm", "gemm", "rope", "p", "o", "act", "vit_out", "final")


def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def _bf2(x: torch.Tensor) -> torch.Tensor:
    hi = _bf(x)
    return hi + _bf(x - hi)


class Rounder:
    def __init__(self, points: Iterable[str] = ALL_POINTS, split: Iterable[str] = ()):
        self.points, self.split = set(points), set(split)
        unknown = (self.points | self.split) - set(ALL_POINTS)
        assert not unknown, unknown

    def __call__(self, name: str, x: torch.Tensor) -> torch.Tensor:
        if name in self.split:
            return _bf2(x)
        return _bf(x) if name in self.points else x


def _attention(q, k, v, mask, hd, R: Rounder):
    """q (H, S, hd), k/v (H, S, hd) already repeated; mask (S, S) bool or None.  Flash form: P = exp(s - rowmax) is the
    bf16 PV operand, the row sum is taken from the fp32 values, O is normalised at the end."""
    s = q @ k.transpose(1, 2) / math.sqrt(hd)
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
    m = s.max(dim=-1, keepdim=True).values
    p = torch.exp(s - m)
    l = p.sum(-1, keepdim=True)
    return (R("p", p) @ v) / l


def vit_forward(w, cfg, pixel_rows, grid_thw, R: Rounder, collect: Optional[list] = None):
    D, Hh = cfg["vit_dim"], cfg["vit_heads"]
    hd, mu = D // Hh, cfg["merge"] ** 2
    q25 = cfg.get("vit_kind", "qwen2") == "qwen2_5"
    x = pixel_rows.float() @ w["visual.patch_embed.proj.weight"].float().reshape(D, -1).t()
    cos, sin = O.vit_rope_tables(grid_thw, hd, cfg["merge"])
    frame_lens = O.vit_segments(grid_thw)
    win = None
    if q25:
        win, win_lens = O.vit_window_index(grid_thw, cfg)
        rows = (win[:, None] * mu + torch.arange(mu)[None, :]).reshape(-1)
        x, cos, sin = x[rows], cos[rows], sin[rows]
    cos, sin = cos[:, None, :], sin[:, None, :]

    def norm(z, p, which):
        if q25:
            return O.rms_norm(z, w[p + which + ".weight"], 1e-6)
        return O.layer_norm(z, w[p + which + ".weight"], w[p + which + ".bias"])

    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        x_in = x
        h = h1 = R("norm", norm(x, p, "norm1"))
        qkv = R("gemm", h @ w[p + "attn.qkv.weight"].float().t() + w[p + "attn.qkv.bias"].float())
        q, k, v = qkv.view(-1, 3, Hh, hd).unbind(1)
        q = R("rope", q * cos + O._rot_half(q) * sin)
        k = R("rope", k * cos + O._rot_half(k) * sin)
        segs = frame_lens if (not q25 or i in cfg["vit_fullatt"]) else win_lens
        outs, s0 = [], 0
        for L in segs:
            qs, ks, vs = (z[s0:s0 + L].transpose(0, 1) for z in (q, k, v))
            outs.append(_attention(qs, ks, vs, None, hd, R).transpose(0, 1).reshape(L, D))
            s0 += L
        a = a_o = R("o", torch.cat(outs, 0))
        x = x + a @ w[p + "attn.proj.weight"].float().t() + w[p + "attn.proj.bias"].float()
        h = R("norm", norm(x, p, "norm2"))
        if q25:
            g = R("gemm", h @ w[p + "mlp.gate_proj.weight"].float().t() + w[p + "mlp.gate_proj.bias"].float())
            u = R("gemm", h @ w[p + "mlp.up_proj.weight"].float().t() + w[p + "mlp.up_proj.bias"].float())
            a = R("act", F.silu(g) * u)
            x = x + a @ w[p + "mlp.down_proj.weight"].float().t() + w[p + "mlp.down_proj.bias"].float()
        else:
            f1 = R("gemm", h @ w[p + "mlp.fc1.weight"].float().t() + w[p + "mlp.fc1.bias"].float())
            a = R("act", O.quick_gelu(f1))
            if collect is not None:
                collect.append(dict(x_in=x_in, h=h1, qkv=torch.cat([q.reshape(-1, D), k.reshape(-1, D), v.reshape(-1, D)], 1), o=a_o,
                                    x_mid=x, h2=h, f1=f1, a=a))
            x = x + a @ w[p + "mlp.fc2.weight"].float().t() + w[p + "mlp.fc2.bias"].float()
    if q25:
        h = R("norm", O.rms_norm(x, w["visual.merger.ln_q.weight"], 1e-6)).reshape(-1, mu * D)
    else:
        h = R("norm", O.layer_norm(x, w["visual.merger.ln_q.weight"], w["visual.merger.ln_q.bias"])).reshape(-1, mu * D)
    m1 = R("gemm", h @ w["visual.merger.mlp.0.weight"].float().t() + w["visual.merger.mlp.0.bias"].float())
    g = R("act", O.gelu_erf(m1))
    out = R("vit_out", g @ w["visual.merger.mlp.2.weight"].float().t() + w["visual.merger.mlp.2.bias"].float())
    return out[torch.argsort(win)] if q25 else out


def llm_hidden(w, cfg, embeds, pos3, R: Rounder, attn_mask: Optional[torch.Tensor] = None, collect: Optional[list] = None):
    """fp32 residual stream out of the last decoder layer (before the final norm)."""
    S = embeds.shape[0]
    H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    if attn_mask is None:
        attn_mask = torch.ones(S, S, dtype=torch.bool).tril()
    cos, sin = O.mrope_tables(pos3, cfg)
    cos, sin = cos[:, None, :], sin[:, None, :]
    x = embeds.float()
    rep = H // KV
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        x_in = x
        h = h1 = R("norm", O.rms_norm(x, w[p + "input_layernorm.weight"], cfg["rms_eps"]))
        q = R("gemm", h @ w[p + "self_attn.q_proj.weight"].float().t() + w[p + "self_attn.q_proj.bias"].float()).view(S, H, hd)
        k = R("gemm", h @ w[p + "self_attn.k_proj.weight"].float().t() + w[p + "self_attn.k_proj.bias"].float()).view(S, KV, hd)
        v = R("gemm", h @ w[p + "self_attn.v_proj.weight"].float().t() + w[p + "self_attn.v_proj.bias"].float()).view(S, KV, hd)
        q = R("rope", q * cos + O._rot_half(q) * sin)
        k = R("rope", k * cos + O._rot_half(k) * sin)
        kk = k.repeat_interleave(rep, dim=1).transpose(0, 1)
        vv = v.repeat_interleave(rep, dim=1).transpose(0, 1)
        a = a_o = R("o", _attention(q.transpose(0, 1), kk, vv, attn_mask, hd, R).transpose(0, 1).reshape(S, H * hd))
        x = x_mid = x + a @ w[p + "self_attn.o_proj.weight"].float().t()
        h = R("norm", O.rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg["rms_eps"]))
        g = R("gemm", h @ w[p + "mlp.gate_proj.weight"].float().t())
        u = R("gemm", h @ w[p + "mlp.up_proj.weight"].float().t())
        a = R("act", F.silu(g) * u)
        if collect is not None:
            collect.append(dict(x_in=x_in, h=h1, qkv=torch.cat([q.reshape(S, -1), k.reshape(S, -1), v.reshape(S, -1)], 1), o=a_o,
                                x_mid=x_mid, h2=h, gu=torch.cat([g, u], 1), a=a))
        x = x + a @ w[p + "mlp.down_proj.weight"].float().t()
    return x


def completion_logps(w, cfg, prompt_ids, completion_ids, pixel_rows, grid_thw, *, points=ALL_POINTS, split=(),
                     video_embeds=None):
    """Same slice as oracle.completion_logps.  ``points`` / ``split``: see the module docstring."""
    R = Rounder(points, split)
    P = prompt_ids.numel()
    ve = video_embeds
    if ve is None and pixel_rows is not None:
        ve = vit_forward(w, cfg, pixel_rows, grid_thw, R)
    rows = []
    for comp in completion_ids:
        ids = torch.cat([prompt_ids, comp])
        e = O.embed_with_video(w, cfg, ids, ve)
        pos3, _ = O.mrope_position_ids(ids.tolist(), grid_thw or [], cfg)
        x = llm_hidden(w, cfg, e, pos3, R)
        hn = R("final", O.rms_norm(x[P - 1:-1], w["model.norm.weight"], cfg["rms_eps"]))
        lg = hn @ O.lm_head_weight(w, cfg).float().t()
        lp = torch.log_softmax(lg, dim=-1)
        rows.append(lp.gather(1, comp.unsqueeze(1)).squeeze(1))
    return torch.stack(rows)
