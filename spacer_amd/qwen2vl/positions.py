"""Host-side integer code: M-RoPE position ids and the rotary tables the kernels consume.

Behaviour follows HF ``Qwen2VLModel.get_rope_index`` (the routine behind the reference's
``model(input_ids, ...)`` / ``generate`` calls, SG_RLVR_trainer.py:357,463): text tokens advance one position
on all three axes; a run of vision placeholders starting at running position s gets (t, h, w) from a
(gt, gh/m, gw/m) mesh offset by s; the next text position is s + max(gh, gw)/m (``era_rule=False``,
transformers 5.x) or max(vision positions)+1 (``era_rule=True``, the 4.x rule of the reference's era).
The two coincide whenever gt <= max(gh, gw)/m.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch

from .config import Qwen2VLConfig


def mrope_positions(ids: Sequence[int], grids: Sequence[Tuple[int, int, int]], cfg: Qwen2VLConfig,
                    era_rule: bool = False, second_per_grid_ts: Sequence[float] = None) -> Tuple[torch.Tensor, int]:
    """Returns (pos [3, S] int64 on CPU, rope_delta) for one unpadded sequence.

    Qwen2.5-VL (cfg.vit_kind == "qwen2_5") spaces the temporal positions of a VIDEO run by
    tokens_per_second * second_per_grid_t (HF modeling_qwen2_5_vl.py:1023-1030; default 1 s per grid step -- what the
    reference's scoring forward uses, it deletes the processor's value at SG_RLVR_trainer.py:519-520): transformers 5.x
    truncates the seconds to an int before multiplying, the 4.x era floors the product per step (``era_rule``)."""
    vis = (cfg.video_token_id, cfg.image_token_id)
    m = cfg.merge
    S = len(ids)
    pos = torch.empty(3, S, dtype=torch.int64)
    cur, i, gi = 0, 0, 0
    while i < S:
        if ids[i] in vis:
            gt, gh, gw = grids[gi]
            gi += 1
            lh, lw = gh // m, gw // m
            n = gt * lh * lw
            if i + n > S or any(t not in vis for t in ids[i:i + n]):
                raise ValueError("vision placeholder run does not match its grid")
            ar = torch.arange(n)
            t_idx = ar // (lh * lw)
            if cfg.vit_kind == "qwen2_5" and ids[i] == cfg.video_token_id:
                sec = 1.0 if second_per_grid_ts is None else float(second_per_grid_ts[gi - 1])
                t_idx = (t_idx.double() * sec * cfg.tokens_per_second).long() if era_rule else t_idx * (cfg.tokens_per_second * int(sec))
            pos[0, i:i + n] = t_idx + cur
            pos[1, i:i + n] = (ar // lw) % lh + cur
            pos[2, i:i + n] = ar % lw + cur
            cur += max(int(t_idx.max()) + 1, lh, lw) if era_rule else max(lh, lw)
            i += n
        else:
            j = i
            while j < S and ids[j] not in vis:
                j += 1
            pos[:, i:j] = torch.arange(cur, cur + (j - i))
            cur += j - i
            i = j
    return pos, int(pos.max()) + 1 - S


def mrope_tables(pos3: torch.Tensor, cfg: Qwen2VLConfig, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin fp32 [S, head_dim]: frequency j takes its angle from row t/h/w by mrope_section, both halves equal."""
    D = cfg.head_dim
    half = D // 2
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=device) / D))
    row = torch.repeat_interleave(torch.arange(3, device=device), torch.tensor(cfg.mrope_section, device=device))
    assert row.numel() == half, "mrope_section must sum to head_dim/2"
    p = pos3.to(device=device, dtype=torch.float32)                       # [3, S]
    ang = p[row, :].t().contiguous() * inv                                # [S, half]
    ang = torch.cat([ang, ang], dim=1)
    return ang.cos().contiguous(), ang.sin().contiguous()


def text_tables(positions: torch.Tensor, cfg: Qwen2VLConfig) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin for pure-text positions (all three rows equal): positions int tensor [S] on the target device."""
    D = cfg.head_dim
    inv = 1.0 / (cfg.rope_theta ** (torch.arange(0, D, 2, dtype=torch.float32, device=positions.device) / D))
    ang = positions.to(torch.float32)[:, None] * inv
    ang = torch.cat([ang, ang], dim=1)
    return ang.cos().contiguous(), ang.sin().contiguous()


def vit_hw_positions(grids: Sequence[Tuple[int, int, int]], merge: int) -> torch.Tensor:
    """[Np, 2] (h, w) of each patch in merge-block-major token order (HF get_vision_position_ids)."""
    out: List[torch.Tensor] = []
    for gt, gh, gw in grids:
        h = torch.arange(gh).view(gh // merge, merge, 1, 1).expand(gh // merge, merge, gw // merge, merge)
        w = torch.arange(gw).view(1, 1, gw // merge, merge).expand(gh // merge, merge, gw // merge, merge)
        hw = torch.stack([h.permute(0, 2, 1, 3).reshape(-1), w.permute(0, 2, 1, 3).reshape(-1)], dim=-1)
        out.append(hw.repeat(gt, 1))
    return torch.cat(out, 0)


def vit_tables(grids, cfg: Qwen2VLConfig, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """cos/sin fp32 [Np, vit_head_dim] of the 2-D vision rotary (theta 10000 over head_dim/2)."""
    hd = cfg.vit_head_dim
    q = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, q, 2, dtype=torch.float32, device=device) / q))
    hw = vit_hw_positions(grids, cfg.merge).to(device=device, dtype=torch.float32)       # [Np, 2]
    ang = (hw[:, :, None] * inv).reshape(hw.shape[0], -1)                                 # [Np, q]
    ang = torch.cat([ang, ang], dim=1)
    return ang.cos().contiguous(), ang.sin().contiguous()


def vit_window_plan(grids, cfg: Qwen2VLConfig):
    """Qwen2.5-VL window attention layout (HF vision_utils.get_vision_window_index, modeling_qwen2_5_vl.py:425-466).

    Merge units (m x m patches) of every temporal grid step are regrouped so that each (window/m/patch)^2-unit window is
    contiguous; border windows are ragged.  Returns
      unit_perm  int64 [Nu]: permuted position u holds original merge unit unit_perm[u]
      row_perm   int64 [Np]: the same at patch-row granularity (m*m rows per unit)
      win_segs   attention segments (start, len, 0, 0) of the windows, in permuted patch rows
    Frames stay contiguous under the permutation, so the full-attention blocks use vit_segments(grids) unchanged."""
    m, mu = cfg.merge, cfg.merge ** 2
    ws = cfg.vit_window // cfg.merge // cfg.patch
    perm: List[torch.Tensor] = []
    segs: List[Tuple[int, int, int, int]] = []
    base, row = 0, 0
    for gt, gh, gw in grids:
        lh, lw = gh // m, gw // m
        idx = torch.arange(gt * lh * lw).reshape(gt, lh, lw)
        ph, pw = ws - lh % ws, ws - lw % ws                    # HF pads a whole extra window when divisible (empty windows)
        nh, nw = (lh + ph) // ws, (lw + pw) // ws
        pad = torch.nn.functional.pad(idx, (0, pw, 0, ph), "constant", -100)
        pad = pad.reshape(gt, nh, ws, nw, ws).permute(0, 1, 3, 2, 4).reshape(gt, nh * nw, ws, ws)
        lens = (pad != -100).sum([2, 3]).reshape(-1)
        flat = pad.reshape(-1)
        perm.append(flat[flat != -100] + base)
        for n in lens.tolist():
            if n:
                segs.append((row, n * mu, 0, 0))
                row += n * mu
        base += gt * lh * lw
    unit_perm = torch.cat(perm)
    row_perm = (unit_perm[:, None] * mu + torch.arange(mu)[None, :]).reshape(-1)
    return unit_perm, row_perm, segs


def vit_segments(grids) -> List[Tuple[int, int, int, int]]:
    """One non-causal attention segment per temporal grid step (HF vision cu_seqlens)."""
    segs, s = [], 0
    for gt, gh, gw in grids:
        for _ in range(gt):
            segs.append((s, gh * gw, 0, 0))
            s += gh * gw
    return segs
