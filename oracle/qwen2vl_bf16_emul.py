"""CPU emulation of the REFERENCE'S OWN numerics: the same arithmetic as oracle/qwen2vl_fp32.py, but every operator
output is rounded to bfloat16 the way HF's eager bf16 path (``--bf16`` + ``torch_dtype=bfloat16``, SG_RLVR_trainer.py:163-190)
produces bf16 tensors after each op -- including the residual stream.  TEST INFRASTRUCTURE ONLY.

Purpose: the north-star asks for log-probs "within 1e-3 of reference".  Against the fp32 oracle no bf16 pipeline can
promise that end to end (each bf16 operand carries 2^-9 relative rounding); what CAN be checked is that this engine
(bf16 operands, fp32 accumulation, fp32 residual stream) is at least as close to the fp32 truth as the reference's
bf16 eager path is.  tests/test_engine_gpu.py asserts err(engine, fp32) <= err(this emulation, fp32) + 1e-3.
Norm statistics / softmax / rotary are computed in fp32 and rounded on output, as HF does.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import qwen2vl_fp32 as O


def r(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).float()


def _vit_forward_qwen2_5(w, cfg, pixel_rows, grid_thw):
    """bf16-eager emulation of oracle.vit_forward_qwen2_5 (HF Qwen2.5-VL vision tower)."""
    D, Hh = cfg["vit_dim"], cfg["vit_heads"]
    hd, mu = D // Hh, cfg["merge"] ** 2
    x = r(r(pixel_rows.float()) @ w["visual.patch_embed.proj.weight"].float().reshape(D, -1).t())
    win, win_lens = O.vit_window_index(grid_thw, cfg)
    rows = (win[:, None] * mu + torch.arange(mu)[None, :]).reshape(-1)
    x = x[rows]
    cos, sin = O.vit_rope_tables(grid_thw, hd, cfg["merge"])
    cos, sin = cos[rows][:, None, :], sin[rows][:, None, :]
    frame_lens = O.vit_segments(grid_thw)

    def rms(z, wt):
        return r(wt.float() * r(z * torch.rsqrt(z.pow(2).mean(-1, keepdim=True) + 1e-6)))

    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        h = rms(x, w[p + "norm1.weight"])
        qkv = r(h @ w[p + "attn.qkv.weight"].float().t() + w[p + "attn.qkv.bias"].float())
        q, k, v = qkv.view(-1, 3, Hh, hd).unbind(1)
        q = r(q * cos + O._rot_half(q) * sin)
        k = r(k * cos + O._rot_half(k) * sin)
        outs, s0 = [], 0
        for L in (frame_lens if i in cfg["vit_fullatt"] else win_lens):
            qs, ks, vs = (z[s0:s0 + L].transpose(0, 1) for z in (q, k, v))
            a = r(torch.softmax(r(qs @ ks.transpose(1, 2)) / math.sqrt(hd), dim=-1))
            outs.append(r(a @ vs).transpose(0, 1).reshape(L, D))
            s0 += L
        a = torch.cat(outs, 0)
        x = r(x + r(a @ w[p + "attn.proj.weight"].float().t() + w[p + "attn.proj.bias"].float()))
        h = rms(x, w[p + "norm2.weight"])
        g = r(h @ w[p + "mlp.gate_proj.weight"].float().t() + w[p + "mlp.gate_proj.bias"].float())
        u = r(h @ w[p + "mlp.up_proj.weight"].float().t() + w[p + "mlp.up_proj.bias"].float())
        x = r(x + r(r(r(F.silu(g)) * u) @ w[p + "mlp.down_proj.weight"].float().t() + w[p + "mlp.down_proj.bias"].float()))
    h = rms(x, w["visual.merger.ln_q.weight"]).reshape(-1, mu * D)
    h = r(O.gelu_erf(r(h @ w["visual.merger.mlp.0.weight"].float().t() + w["visual.merger.mlp.0.bias"].float())))
    out = r(h @ w["visual.merger.mlp.2.weight"].float().t() + w["visual.merger.mlp.2.bias"].float())
    return out[torch.argsort(win)]


def vit_forward(w, cfg, pixel_rows, grid_thw):
    if cfg.get("vit_kind", "qwen2") == "qwen2_5":
        return _vit_forward_qwen2_5(w, cfg, pixel_rows, grid_thw)
    D, Hh = cfg["vit_dim"], cfg["vit_heads"]
    hd = D // Hh
    x = r(r(pixel_rows.float()) @ w["visual.patch_embed.proj.weight"].float().reshape(D, -1).t())
    cos, sin = O.vit_rope_tables(grid_thw, hd, cfg["merge"])
    cos, sin = cos[:, None, :], sin[:, None, :]
    segs = O.vit_segments(grid_thw)
    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        h = r(O.layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"]))
        qkv = r(h @ w[p + "attn.qkv.weight"].float().t() + w[p + "attn.qkv.bias"].float())
        q, k, v = qkv.view(-1, 3, Hh, hd).unbind(1)
        q = r(q * cos + O._rot_half(q) * sin)
        k = r(k * cos + O._rot_half(k) * sin)
        outs, s0 = [], 0
        for L in segs:
            qs, ks, vs = (z[s0:s0 + L].transpose(0, 1) for z in (q, k, v))
            a = r(torch.softmax(r(qs @ ks.transpose(1, 2)) / math.sqrt(hd), dim=-1))
            outs.append(r(a @ vs).transpose(0, 1).reshape(L, D))
            s0 += L
        a = torch.cat(outs, 0)
        x = r(x + r(a @ w[p + "attn.proj.weight"].float().t() + w[p + "attn.proj.bias"].float()))
        h = r(O.layer_norm(x, w[p + "norm2.weight"], w[p + "norm2.bias"]))
        h = r(O.quick_gelu(r(h @ w[p + "mlp.fc1.weight"].float().t() + w[p + "mlp.fc1.bias"].float())))
        x = r(x + r(h @ w[p + "mlp.fc2.weight"].float().t() + w[p + "mlp.fc2.bias"].float()))
    m = cfg["merge"] ** 2
    h = r(O.layer_norm(x, w["visual.merger.ln_q.weight"], w["visual.merger.ln_q.bias"])).reshape(-1, m * D)
    h = r(O.gelu_erf(r(h @ w["visual.merger.mlp.0.weight"].float().t() + w["visual.merger.mlp.0.bias"].float())))
    return r(h @ w["visual.merger.mlp.2.weight"].float().t() + w["visual.merger.mlp.2.bias"].float())


def llm_logits(w, cfg, embeds, pos3):
    S = embeds.shape[0]
    H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    mask = torch.ones(S, S, dtype=torch.bool).tril()
    cos, sin = O.mrope_tables(pos3, cfg)
    cos, sin = r(cos)[:, None, :], r(sin)[:, None, :]          # HF casts cos/sin to the model dtype
    x = r(embeds.float())
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        h = r(w[p + "input_layernorm.weight"].float() * r(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg["rms_eps"])))
        q = r(h @ w[p + "self_attn.q_proj.weight"].float().t() + w[p + "self_attn.q_proj.bias"].float()).view(S, H, hd)
        k = r(h @ w[p + "self_attn.k_proj.weight"].float().t() + w[p + "self_attn.k_proj.bias"].float()).view(S, KV, hd)
        v = r(h @ w[p + "self_attn.v_proj.weight"].float().t() + w[p + "self_attn.v_proj.bias"].float()).view(S, KV, hd)
        q = r(r(q * cos) + r(O._rot_half(q) * sin))
        k = r(r(k * cos) + r(O._rot_half(k) * sin))
        rep = H // KV
        kk = k.repeat_interleave(rep, dim=1).transpose(0, 1)
        vv = v.repeat_interleave(rep, dim=1).transpose(0, 1)
        s = r(r(q.transpose(0, 1) @ kk.transpose(1, 2)) / math.sqrt(hd)).masked_fill(~mask, float("-inf"))
        a = r(r(torch.softmax(s, dim=-1)) @ vv).transpose(0, 1).reshape(S, H * hd)
        x = r(x + r(a @ w[p + "self_attn.o_proj.weight"].float().t()))
        h = r(w[p + "post_attention_layernorm.weight"].float() * r(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg["rms_eps"])))
        g = r(r(F.silu(r(h @ w[p + "mlp.gate_proj.weight"].float().t()))) * r(h @ w[p + "mlp.up_proj.weight"].float().t()))
        x = r(x + r(g @ w[p + "mlp.down_proj.weight"].float().t()))
    x = r(w["model.norm.weight"].float() * r(x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + cfg["rms_eps"])))
    return r(x @ O.lm_head_weight(w, cfg).float().t())          # bf16 logits, as the reference's log_softmax sees them


def completion_logps(w, cfg, prompt_ids, completion_ids, pixel_rows, grid_thw):
    """Same slice as oracle.completion_logps, with log_softmax in bf16-rounded logits (TR:360-365 runs in the logits dtype)."""
    P = prompt_ids.numel()
    ve = vit_forward(w, cfg, pixel_rows, grid_thw) if pixel_rows is not None else None
    rows = []
    for comp in completion_ids:
        ids = torch.cat([prompt_ids, comp])
        e = O.embed_with_video(w, cfg, ids, ve)
        pos3, _ = O.mrope_position_ids(ids.tolist(), grid_thw or [], cfg)
        lg = llm_logits(w, cfg, e, pos3)
        lp = r(torch.log_softmax(lg[:-1], dim=-1))
        rows.append(lp.gather(1, ids[1:].unsqueeze(1)).squeeze(1)[P - 1:])
    return torch.stack(rows)
