"""Seeded synthetic workload of BASELINE.md / SURVEY 8(d): random uint8 frames already at post-smart_resize size,
a prompt of <|vision_start|> Nv x <|video_pad|> <|vision_end|> followed by random text ids, and per-rollout rewards
U[0, 2] (reward functions on decoded text are meaningless for random weights; they are timed on the golden table)."""
from __future__ import annotations

from typing import Tuple

import torch

from . import kernels as K
from .qwen2vl.config import Qwen2VLConfig
from .rollout import PromptInput


def synthetic_frames(prompt_idx: int, frames: int, height: int, width: int, device) -> torch.Tensor:
    g = torch.Generator().manual_seed(1000 + prompt_idx)
    return torch.randint(0, 256, (frames, 3, height, width), generator=g, dtype=torch.uint8).to(device)


def synthetic_prompt_ids(cfg: Qwen2VLConfig, prompt_idx: int, n_video_tokens: int, n_text: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(2000 + prompt_idx)
    hi = min(150000, cfg.vocab)
    lo = min(1000, hi // 2)
    special = {cfg.image_token_id, cfg.video_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_token_id, cfg.pad_token_id}
    text = torch.randint(lo, hi, (n_text,), generator=g)
    for s in special:
        text[text == s] = lo
    return torch.cat([torch.tensor([cfg.vision_start_id]), torch.full((n_video_tokens,), cfg.video_token_id),
                      torch.tensor([cfg.vision_end_id]), text]).long()


def make_prompt(cfg: Qwen2VLConfig, prompt_idx: int, frames: int, height: int, width: int, n_text: int, device,
                frames_u8: torch.Tensor = None) -> Tuple[PromptInput, torch.Tensor]:
    """Returns (PromptInput with patchified pixels, the uint8 frames).  Patchify runs on the GPU (kernel K1)."""
    if frames_u8 is None:
        frames_u8 = synthetic_frames(prompt_idx, frames, height, width, device)
    pix, grid = K.patchify(frames_u8, cfg.patch, cfg.tpatch, cfg.merge, cfg.patch_kpad)
    nv = grid[0] * grid[1] * grid[2] // (cfg.merge ** 2)
    ids = synthetic_prompt_ids(cfg, prompt_idx, nv, n_text).to(device)
    return PromptInput(ids=ids, pix=pix, grids=[tuple(grid)]), frames_u8


def synthetic_rewards(step: int, prompt_idx: int, num_generations: int) -> torch.Tensor:
    """[K, 2] (accuracy in U[0,2]... split as accuracy U[0,1]+map bonus and format {0,1}); seeded per group."""
    g = torch.Generator().manual_seed(3000 + 131 * step + prompt_idx)
    acc = torch.rand(num_generations, generator=g) * 2.0
    acc = torch.where(acc < 0.6, torch.zeros_like(acc), acc)       # some wrong answers, as real groups have
    fmt = (torch.rand(num_generations, generator=g) > 0.3).float()
    return torch.stack([acc, fmt], dim=1)
