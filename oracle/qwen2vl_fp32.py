"""CPU oracle: fp32 restatement of the Qwen2-VL arithmetic the SG-RLVR hot path runs.

TEST INFRASTRUCTURE ONLY.  Nothing under ``spacer_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it,
and only as the checker / the timed CPU baseline.

Provenance.  The reference repository (OuyangKun10/SpaceR) owns no model arithmetic: its
trainer calls ``model.generate`` (SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py:463)
and ``model(input_ids, **kwargs).logits`` (same file :357) on HF ``transformers``
``Qwen2VLForConditionalGeneration`` -- a third-party dependency that is NOT under
/root/reference (pinned only by a commented git hash in r1-v/setup.py:64).  This file
restates that published architecture from the math:

  * ViT: Conv3d patch embed == GEMM over 1176-wide patch rows; 32 x (LayerNorm, fused qkv,
    2-D rotary in fp32, per-temporal-grid non-causal attention, proj, quick_gelu MLP);
    PatchMerger (LayerNorm, 2x2 merge by view, Linear-GELU-Linear).
  * LLM: RMSNorm (fp32), q/k/v with bias, M-RoPE sections [16,24,24], causal GQA attention,
    SwiGLU MLP, final norm, lm_head.
  * per-token log-probs as SG_RLVR_trainer.py:353-366 computes them.

Pinning.  The reference has no tests or golden vectors for this arithmetic ("parity
unpinned" by the reference itself).  The restatement is pinned instead
against HF transformers 5.15.0 run in the authoring container
(scripts/make_golden_model.py -> tests/golden/tiny_model_*.npz; tests/test_oracle_model.py).

Everything here is plain torch on fp32 tensors, written for clarity, not speed.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

Weights = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------
# configuration (plain dict so the oracle does not depend on the product's config class)
# --------------------------------------------------------------------------------------
def make_config(
    *,
    hidden: int,
    layers: int,
    heads: int,
    kv_heads: int,
    intermediate: int,
    vocab: int,
    vit_dim: int,
    vit_depth: int,
    vit_heads: int,
    vit_mlp: int,
    head_dim: Optional[int] = None,
    mrope_section: Sequence[int] = (16, 24, 24),
    rope_theta: float = 1e6,
    rms_eps: float = 1e-6,
    patch: int = 14,
    tpatch: int = 2,
    merge: int = 2,
    tie_embeddings: bool = False,
    video_token_id: int = 151656,
    image_token_id: int = 151655,
    vit_kind: str = "qwen2",
    vit_window: int = 112,
    vit_fullatt: Sequence[int] = (),
    tokens_per_second: int = 2,
) -> dict:
    hd = head_dim or hidden // heads
    return dict(
        hidden=hidden, layers=layers, heads=heads, kv_heads=kv_heads, head_dim=hd,
        intermediate=intermediate, vocab=vocab, vit_dim=vit_dim, vit_depth=vit_depth,
        vit_heads=vit_heads, vit_mlp=vit_mlp, mrope_section=tuple(mrope_section),
        rope_theta=float(rope_theta), rms_eps=float(rms_eps), patch=patch, tpatch=tpatch,
        merge=merge, tie_embeddings=tie_embeddings, video_token_id=video_token_id,
        image_token_id=image_token_id, vit_kind=vit_kind, vit_window=vit_window, vit_fullatt=tuple(vit_fullatt),
        tokens_per_second=tokens_per_second,
    )


# --------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------
def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w.float() * (x.float() * torch.rsqrt(var + eps))


def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    x = x.float()
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w.float() + b.float()


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def gelu_erf(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _rot_half(x: torch.Tensor) -> torch.Tensor:
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


# --------------------------------------------------------------------------------------
# vision tower
# --------------------------------------------------------------------------------------
def vit_position_hw(grid_thw: Sequence[Sequence[int]], merge: int) -> torch.Tensor:
    """(N, 2) integer (h, w) position of every patch in merge-block-major token order."""
    out = []
    for t, h, w in grid_thw:
        hp = torch.arange(h).view(h, 1).expand(h, w)
        wp = torch.arange(w).view(1, w).expand(h, w)

        def blk(z):
            return z.reshape(h // merge, merge, w // merge, merge).permute(0, 2, 1, 3).reshape(-1)

        hw = torch.stack([blk(hp), blk(wp)], dim=-1)
        out.append(hw.repeat(t, 1))
    return torch.cat(out, 0)


def vit_rope_tables(grid_thw, head_dim: int, merge: int, theta: float = 10000.0):
    """cos/sin of shape (N, head_dim): [h*f(0..d/4), w*f(0..d/4)] duplicated over both halves."""
    q = head_dim // 2
    inv = 1.0 / (theta ** (torch.arange(0, q, 2, dtype=torch.float32) / q))
    pos = vit_position_hw(grid_thw, merge).float()            # (N, 2)
    fr = (pos.unsqueeze(-1) * inv).flatten(1)                  # (N, q)
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def vit_segments(grid_thw) -> List[int]:
    """Attention segment lengths: one per temporal grid step (h*w patches each)."""
    seg = []
    for t, h, w in grid_thw:
        seg += [h * w] * t
    return seg


def vit_window_index(grid_thw, cfg: dict):
    """Qwen2.5-VL window regrouping (HF vision_utils.get_vision_window_index, used at modeling_qwen2_5_vl.py:425-431):
    returns (window_index over merge units, window lengths in PATCHES, in permuted order; empty windows dropped)."""
    m = cfg["merge"]
    ws = cfg["vit_window"] // m // cfg["patch"]
    index, lens, base = [], [], 0
    for t, h, w_ in grid_thw:
        lh, lw = h // m, w_ // m
        idx = torch.arange(t * lh * lw).reshape(t, lh, lw)
        ph, pw = ws - lh % ws, ws - lw % ws
        nh, nw = (lh + ph) // ws, (lw + pw) // ws
        pad = F.pad(idx, (0, pw, 0, ph), "constant", -100).reshape(t, nh, ws, nw, ws).permute(0, 1, 3, 2, 4)
        pad = pad.reshape(t, nh * nw, ws, ws)
        lens += [int(n) * m * m for n in (pad != -100).sum([2, 3]).reshape(-1).tolist() if n]
        flat = pad.reshape(-1)
        index.append(flat[flat != -100] + base)
        base += t * lh * lw
    return torch.cat(index), lens


def _vit_attention(q, k, v, segs, hd):
    outs, s0 = [], 0
    for L in segs:
        qs, ks, vs = (z[s0:s0 + L].transpose(0, 1) for z in (q, k, v))      # (heads, L, hd)
        a = torch.softmax(qs @ ks.transpose(1, 2) / math.sqrt(hd), dim=-1)
        outs.append((a @ vs).transpose(0, 1).reshape(L, -1))
        s0 += L
    return torch.cat(outs, 0)


def vit_forward_qwen2_5(w: Weights, cfg: dict, pixel_rows: torch.Tensor, grid_thw, *, return_hidden=False):
    """Qwen2_5_VisionTransformerPretrainedModel.forward (HF modeling_qwen2_5_vl.py:408-472): patch embed, regroup merge
    units into windows, 2-D rotary in the permuted order, RMSNorm + biased SwiGLU blocks with window attention except
    on cfg['vit_fullatt'] (per-frame attention), RMSNorm merger, inverse permutation of the merged rows."""
    D, Hh = cfg["vit_dim"], cfg["vit_heads"]
    hd, mu = D // Hh, cfg["merge"] ** 2
    x = pixel_rows.float() @ w["visual.patch_embed.proj.weight"].float().reshape(D, -1).t()
    win, win_lens = vit_window_index(grid_thw, cfg)
    rows = (win[:, None] * mu + torch.arange(mu)[None, :]).reshape(-1)
    x = x[rows]
    cos, sin = vit_rope_tables(grid_thw, hd, cfg["merge"])
    cos, sin = cos[rows][:, None, :], sin[rows][:, None, :]
    frame_lens = vit_segments(grid_thw)
    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        h = rms_norm(x, w[p + "norm1.weight"], 1e-6)
        qkv = h @ w[p + "attn.qkv.weight"].float().t() + w[p + "attn.qkv.bias"].float()
        q, k, v = qkv.view(-1, 3, Hh, hd).unbind(1)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        a = _vit_attention(q, k, v, frame_lens if i in cfg["vit_fullatt"] else win_lens, hd)
        x = x + a @ w[p + "attn.proj.weight"].float().t() + w[p + "attn.proj.bias"].float()
        h = rms_norm(x, w[p + "norm2.weight"], 1e-6)
        g = h @ w[p + "mlp.gate_proj.weight"].float().t() + w[p + "mlp.gate_proj.bias"].float()
        u = h @ w[p + "mlp.up_proj.weight"].float().t() + w[p + "mlp.up_proj.bias"].float()
        x = x + (F.silu(g) * u) @ w[p + "mlp.down_proj.weight"].float().t() + w[p + "mlp.down_proj.bias"].float()
    h = rms_norm(x, w["visual.merger.ln_q.weight"], 1e-6).reshape(-1, mu * D)
    h = gelu_erf(h @ w["visual.merger.mlp.0.weight"].float().t() + w["visual.merger.mlp.0.bias"].float())
    out = h @ w["visual.merger.mlp.2.weight"].float().t() + w["visual.merger.mlp.2.bias"].float()
    out = out[torch.argsort(win)]
    return (out, x) if return_hidden else out


def vit_forward(w: Weights, cfg: dict, pixel_rows: torch.Tensor, grid_thw, *, return_hidden=False):
    """pixel_rows: (Np, 3*tpatch*patch*patch) already normalised; returns (Np/merge^2, hidden)."""
    if cfg.get("vit_kind", "qwen2") == "qwen2_5":
        return vit_forward_qwen2_5(w, cfg, pixel_rows, grid_thw, return_hidden=return_hidden)
    D, Hh = cfg["vit_dim"], cfg["vit_heads"]
    hd = D // Hh
    x = pixel_rows.float() @ w["visual.patch_embed.proj.weight"].float().reshape(D, -1).t()
    cos, sin = vit_rope_tables(grid_thw, hd, cfg["merge"])
    cos, sin = cos[:, None, :], sin[:, None, :]
    segs = vit_segments(grid_thw)
    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        h = layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"])
        qkv = h @ w[p + "attn.qkv.weight"].float().t() + w[p + "attn.qkv.bias"].float()
        q, k, v = qkv.view(-1, 3, Hh, hd).unbind(1)                         # (N, heads, hd)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        outs, s0 = [], 0
        for L in segs:
            qs, ks, vs = (z[s0:s0 + L].transpose(0, 1) for z in (q, k, v))  # (heads, L, hd)
            a = torch.softmax(qs @ ks.transpose(1, 2) / math.sqrt(hd), dim=-1)
            outs.append((a @ vs).transpose(0, 1).reshape(L, D))
            s0 += L
        a = torch.cat(outs, 0)
        x = x + a @ w[p + "attn.proj.weight"].float().t() + w[p + "attn.proj.bias"].float()
        h = layer_norm(x, w[p + "norm2.weight"], w[p + "norm2.bias"])
        h = quick_gelu(h @ w[p + "mlp.fc1.weight"].float().t() + w[p + "mlp.fc1.bias"].float())
        x = x + h @ w[p + "mlp.fc2.weight"].float().t() + w[p + "mlp.fc2.bias"].float()
    m = cfg["merge"] ** 2
    h = layer_norm(x, w["visual.merger.ln_q.weight"], w["visual.merger.ln_q.bias"]).reshape(-1, m * D)
    h = gelu_erf(h @ w["visual.merger.mlp.0.weight"].float().t() + w["visual.merger.mlp.0.bias"].float())
    out = h @ w["visual.merger.mlp.2.weight"].float().t() + w["visual.merger.mlp.2.bias"].float()
    return (out, x) if return_hidden else out


# --------------------------------------------------------------------------------------
# M-RoPE position ids
# --------------------------------------------------------------------------------------
def mrope_position_ids(input_ids: Sequence[int], grid_thw, cfg: dict, *, era_rule: bool = False,
                       second_per_grid_ts: Optional[Sequence[float]] = None):
    """(3, S) t/h/w positions and rope_delta for ONE unpadded sequence.

    A run of vision placeholder tokens starting at running position s gets
    t = s + arange(gt), h = s + arange(gh/m), w = s + arange(gw/m) over a (gt, gh/m, gw/m) grid.
    The next text position is s + max(gh, gw)/m (transformers 5.x) or, with ``era_rule``,
    max(all vision positions)+1 (transformers 4.x, the reference's era).  The two agree
    whenever gt <= max(gh, gw)/m  (SURVEY.md 2.3 drift warning).
    Qwen2.5-VL video runs: temporal index * tokens_per_second * second_per_grid_t (default 1 s): transformers 5.x
    multiplies by tps * int(seconds) (modeling_qwen2_5_vl.py:1023-1030), the 4.x era floors index * seconds * tps.
    """
    ids = list(int(i) for i in input_ids)
    vis = {cfg["video_token_id"], cfg["image_token_id"]}
    m = cfg["merge"]
    grids = iter(grid_thw)
    pos, cur, i, n_vis = [], 0, 0, 0
    while i < len(ids):
        if ids[i] in vis:
            gt, gh, gw = next(grids)
            lh, lw = gh // m, gw // m
            n = gt * lh * lw
            assert all(t in vis for t in ids[i:i + n]), "placeholder run shorter than grid"
            ti = torch.arange(gt)
            if cfg.get("vit_kind", "qwen2") == "qwen2_5" and ids[i] == cfg["video_token_id"]:
                sec = 1.0 if second_per_grid_ts is None else float(second_per_grid_ts[n_vis])
                ti = (ti.double() * sec * cfg["tokens_per_second"]).long() if era_rule else ti * (cfg["tokens_per_second"] * int(sec))
            n_vis += 1
            tt = ti.view(gt, 1, 1).expand(gt, lh, lw).reshape(-1) + cur
            hh = torch.arange(lh).view(1, lh, 1).expand(gt, lh, lw).reshape(-1) + cur
            ww = torch.arange(lw).view(1, 1, lw).expand(gt, lh, lw).reshape(-1) + cur
            pos.append(torch.stack([tt, hh, ww]))
            cur = cur + (max(int(ti.max()) + 1, lh, lw) if era_rule else max(lh, lw))
            i += n
        else:
            j = i
            while j < len(ids) and ids[j] not in vis:
                j += 1
            pos.append((torch.arange(j - i) + cur).view(1, -1).expand(3, -1))
            cur += j - i
            i = j
    p = torch.cat(pos, dim=1).long()
    delta = int(p.max()) + 1 - len(ids)
    return p, delta


def mrope_tables(pos3: torch.Tensor, cfg: dict):
    """cos/sin (S, head_dim) with the [16,24,24]x2 section interleave of rows t/h/w."""
    hd = cfg["head_dim"]
    inv = 1.0 / (cfg["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    fr = pos3.float().unsqueeze(-1) * inv                       # (3, S, hd/2)
    emb = torch.cat([fr, fr], dim=-1)                           # (3, S, hd)
    sec = list(cfg["mrope_section"]) * 2
    cos = torch.cat([c[i % 3] for i, c in enumerate(emb.cos().split(sec, dim=-1))], dim=-1)
    sin = torch.cat([s[i % 3] for i, s in enumerate(emb.sin().split(sec, dim=-1))], dim=-1)
    return cos, sin


# --------------------------------------------------------------------------------------
# language model
# --------------------------------------------------------------------------------------
def llm_forward(w: Weights, cfg: dict, embeds: torch.Tensor, pos3: torch.Tensor,
                attn_mask: Optional[torch.Tensor] = None, *, return_hidden: bool = False,
                collect: Optional[dict] = None, final_norm: bool = True) -> torch.Tensor:
    """embeds (S, hidden) for ONE sequence (or a packed layout with ``attn_mask``).

    ``attn_mask`` is an (S, S) boolean "may attend" matrix; default = causal.
    Returns logits (S, vocab) unless ``return_hidden``.
    """
    S = embeds.shape[0]
    H, KV, hd = cfg["heads"], cfg["kv_heads"], cfg["head_dim"]
    if attn_mask is None:
        attn_mask = torch.ones(S, S, dtype=torch.bool).tril()
    cos, sin = mrope_tables(pos3, cfg)
    cos, sin = cos[:, None, :], sin[:, None, :]
    x = embeds.float()
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        h = rms_norm(x, w[p + "input_layernorm.weight"], cfg["rms_eps"])
        q = (h @ w[p + "self_attn.q_proj.weight"].float().t() + w[p + "self_attn.q_proj.bias"].float()).view(S, H, hd)
        k = (h @ w[p + "self_attn.k_proj.weight"].float().t() + w[p + "self_attn.k_proj.bias"].float()).view(S, KV, hd)
        v = (h @ w[p + "self_attn.v_proj.weight"].float().t() + w[p + "self_attn.v_proj.bias"].float()).view(S, KV, hd)
        q = q * cos + _rot_half(q) * sin
        k = k * cos + _rot_half(k) * sin
        if collect is not None:
            collect.setdefault("k", []).append(k.clone())
            collect.setdefault("v", []).append(v.clone())
        rep = H // KV
        kk = k.repeat_interleave(rep, dim=1).transpose(0, 1)     # (H, S, hd)
        vv = v.repeat_interleave(rep, dim=1).transpose(0, 1)
        s = q.transpose(0, 1) @ kk.transpose(1, 2) / math.sqrt(hd)
        s = s.masked_fill(~attn_mask, float("-inf"))
        a = (torch.softmax(s, dim=-1) @ vv).transpose(0, 1).reshape(S, H * hd)
        x = x + a @ w[p + "self_attn.o_proj.weight"].float().t()
        h = rms_norm(x, w[p + "post_attention_layernorm.weight"], cfg["rms_eps"])
        g = F.silu(h @ w[p + "mlp.gate_proj.weight"].float().t()) * (h @ w[p + "mlp.up_proj.weight"].float().t())
        x = x + g @ w[p + "mlp.down_proj.weight"].float().t()
    if not final_norm:                 # layer-local parity tests: the residual stream itself (tests/test_layer_local_gpu.py)
        return x
    x = rms_norm(x, w["model.norm.weight"], cfg["rms_eps"])
    if return_hidden:
        return x
    return x @ lm_head_weight(w, cfg).float().t()


def lm_head_weight(w: Weights, cfg: dict) -> torch.Tensor:
    return w["model.embed_tokens.weight"] if cfg["tie_embeddings"] else w["lm_head.weight"]


def embed_with_video(w: Weights, cfg: dict, input_ids: torch.Tensor, video_embeds: Optional[torch.Tensor]):
    """Token embedding gather + masked scatter of ViT outputs at placeholder ids (HF :1170-1176)."""
    e = w["model.embed_tokens.weight"].float()[input_ids]
    if video_embeds is not None:
        mask = (input_ids == cfg["video_token_id"]) | (input_ids == cfg["image_token_id"])
        assert int(mask.sum()) == video_embeds.shape[0]
        e = e.clone()
        e[mask] = video_embeds.float()
    return e


def full_logits(w: Weights, cfg: dict, input_ids: torch.Tensor, pixel_rows, grid_thw, *, era_rule=False):
    """One unpadded sequence -> (S, vocab) logits, the quantity SG_RLVR_trainer.py:357 reads."""
    ve = vit_forward(w, cfg, pixel_rows, grid_thw) if pixel_rows is not None else None
    e = embed_with_video(w, cfg, input_ids, ve)
    pos3, _ = mrope_position_ids(input_ids.tolist(), grid_thw or [], cfg, era_rule=era_rule)
    return llm_forward(w, cfg, e, pos3)


def per_token_logps(logits: torch.Tensor, input_ids: torch.Tensor) -> torch.Tensor:
    """SG_RLVR_trainer.py:353-366 for one row: log_softmax(logits[:-1])[ids[1:]]  -> (S-1,)."""
    lp = torch.log_softmax(logits[:-1].float(), dim=-1)
    return lp.gather(1, input_ids[1:].unsqueeze(1)).squeeze(1)


def completion_logps(w: Weights, cfg: dict, prompt_ids: torch.Tensor, completion_ids: torch.Tensor,
                     pixel_rows, grid_thw) -> torch.Tensor:
    """(K, C) per-token log-probs of K completions of one prompt, exactly the slice
    ``per_token_logps[:, prompt_length-1:]`` of SG_RLVR_trainer.py:527-528 (each of the K rows is
    an independent causal sequence prompt+completion_k, no padding mask, pads included)."""
    P = prompt_ids.numel()
    ve = vit_forward(w, cfg, pixel_rows, grid_thw) if pixel_rows is not None else None
    rows = []
    for comp in completion_ids:
        ids = torch.cat([prompt_ids, comp])
        e = embed_with_video(w, cfg, ids, ve)
        pos3, _ = mrope_position_ids(ids.tolist(), grid_thw or [], cfg)
        lg = llm_forward(w, cfg, e, pos3)
        rows.append(per_token_logps(lg, ids)[P - 1:])
    return torch.stack(rows)


# --------------------------------------------------------------------------------------
# synthetic weights (seed 1234, N(0, 0.02), norm weight 1 / bias 0)
# --------------------------------------------------------------------------------------
def random_weights(cfg: dict, seed: int = 1234, dtype=torch.float32, std: float = 0.02) -> Weights:
    g = torch.Generator().manual_seed(seed)

    def n(*shape):
        return (torch.randn(*shape, generator=g) * std).to(dtype)

    D, Hd, I, V = cfg["vit_dim"], cfg["hidden"], cfg["intermediate"], cfg["vocab"]
    hd, H, KV = cfg["head_dim"], cfg["heads"], cfg["kv_heads"]
    pk = 3 * cfg["tpatch"] * cfg["patch"] ** 2
    w: Weights = {"visual.patch_embed.proj.weight": n(D, pk)}
    for i in range(cfg["vit_depth"]):
        p = f"visual.blocks.{i}."
        w[p + "norm1.weight"] = torch.ones(D, dtype=dtype); w[p + "norm1.bias"] = torch.zeros(D, dtype=dtype)
        w[p + "norm2.weight"] = torch.ones(D, dtype=dtype); w[p + "norm2.bias"] = torch.zeros(D, dtype=dtype)
        w[p + "attn.qkv.weight"] = n(3 * D, D); w[p + "attn.qkv.bias"] = n(3 * D)
        w[p + "attn.proj.weight"] = n(D, D); w[p + "attn.proj.bias"] = n(D)
        if cfg.get("vit_kind", "qwen2") == "qwen2_5":
            del w[p + "norm1.bias"], w[p + "norm2.bias"]
            for nm in ("gate_proj", "up_proj"):
                w[p + f"mlp.{nm}.weight"] = n(cfg["vit_mlp"], D); w[p + f"mlp.{nm}.bias"] = n(cfg["vit_mlp"])
            w[p + "mlp.down_proj.weight"] = n(D, cfg["vit_mlp"]); w[p + "mlp.down_proj.bias"] = n(D)
            continue
        w[p + "mlp.fc1.weight"] = n(cfg["vit_mlp"], D); w[p + "mlp.fc1.bias"] = n(cfg["vit_mlp"])
        w[p + "mlp.fc2.weight"] = n(D, cfg["vit_mlp"]); w[p + "mlp.fc2.bias"] = n(D)
    m = cfg["merge"] ** 2
    w["visual.merger.ln_q.weight"] = torch.ones(D, dtype=dtype); w["visual.merger.ln_q.bias"] = torch.zeros(D, dtype=dtype)
    if cfg.get("vit_kind", "qwen2") == "qwen2_5":
        del w["visual.merger.ln_q.bias"]
    w["visual.merger.mlp.0.weight"] = n(m * D, m * D); w["visual.merger.mlp.0.bias"] = n(m * D)
    w["visual.merger.mlp.2.weight"] = n(Hd, m * D); w["visual.merger.mlp.2.bias"] = n(Hd)
    w["model.embed_tokens.weight"] = n(V, Hd)
    for i in range(cfg["layers"]):
        p = f"model.layers.{i}."
        w[p + "input_layernorm.weight"] = torch.ones(Hd, dtype=dtype)
        w[p + "post_attention_layernorm.weight"] = torch.ones(Hd, dtype=dtype)
        w[p + "self_attn.q_proj.weight"] = n(H * hd, Hd); w[p + "self_attn.q_proj.bias"] = n(H * hd)
        w[p + "self_attn.k_proj.weight"] = n(KV * hd, Hd); w[p + "self_attn.k_proj.bias"] = n(KV * hd)
        w[p + "self_attn.v_proj.weight"] = n(KV * hd, Hd); w[p + "self_attn.v_proj.bias"] = n(KV * hd)
        w[p + "self_attn.o_proj.weight"] = n(Hd, H * hd)
        w[p + "mlp.gate_proj.weight"] = n(I, Hd); w[p + "mlp.up_proj.weight"] = n(I, Hd)
        w[p + "mlp.down_proj.weight"] = n(Hd, I)
    w["model.norm.weight"] = torch.ones(Hd, dtype=dtype)
    if not cfg["tie_embeddings"]:
        w["lm_head.weight"] = n(V, Hd)
    return w


# --------------------------------------------------------------------------------------
# frame -> patch rows (HF video processor: rescale 1/255, CLIP mean/std, patchify permute)
# --------------------------------------------------------------------------------------
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def patchify_frames(frames_u8: torch.Tensor, cfg: dict):
    """frames (F, 3, H, W) uint8 (H, W multiples of patch*merge) -> (Np, 1176) f32, grid (gt, gh, gw).

    Token order is merge-block-major; feature order is (c, tp, py, px) fastest-last; an odd frame
    count is padded by repeating the last frame.
    """
    ps, tp, m = cfg["patch"], cfg["tpatch"], cfg["merge"]
    x = frames_u8.float() / 255.0
    mean = torch.tensor(CLIP_MEAN).view(1, 3, 1, 1)
    std = torch.tensor(CLIP_STD).view(1, 3, 1, 1)
    x = (x - mean) / std
    if x.shape[0] % tp:
        x = torch.cat([x, x[-1:].expand(tp - x.shape[0] % tp, -1, -1, -1)], 0)
    Fr, C, H, W = x.shape
    gt, gh, gw = Fr // tp, H // ps, W // ps
    x = x.view(gt, tp, C, gh // m, m, ps, gw // m, m, ps)
    x = x.permute(0, 3, 6, 4, 7, 2, 1, 5, 8)        # gt, gh/m, gw/m, m, m, C, tp, ps, ps
    return x.reshape(gt * gh * gw, C * tp * ps * ps).contiguous(), (gt, gh, gw)
