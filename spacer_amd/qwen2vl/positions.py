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
                    era_rule: bool = False) -> Tuple[torch.Tensor, int]:
    """Returns (pos [3, S] int64 on CPU, rope_delta) for one unpadded sequence."""
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
            pos[0, i:i + n] = ar // (lh * lw) + cur
            pos[1, i:i + n] = (ar // lw) % lh + cur
            pos[2, i:i + n] = ar % lw + cur
            cur += max(gt, lh, lw) if era_rule else max(lh, lw)
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


def vit_segments(grids) -> List[Tuple[int, int, int, int]]:
    """One non-causal attention segment per temporal grid step (HF vision cu_seqlens)."""
    segs, s = [], 0
    for gt, gh, gw in grids:
        for _ in range(gt):
            segs.append((s, gh * gw, 0, 0))
            s += gh * gw
    return segs
