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


# ------------------------------------------------------------------------------------- a processor for synthetic text
class SyntheticTokenizer:
    """``processor.tokenizer`` surface over whitespace "words": special markers map to the model's special ids (one id per
    <|video_pad|> / <|image_pad|>), every other word hashes into the ordinary vocabulary."""

    def __init__(self, cfg: Qwen2VLConfig):
        self.cfg = cfg
        self.special = {"<|vision_start|>": cfg.vision_start_id, "<|vision_end|>": cfg.vision_end_id,
                        "<|video_pad|>": cfg.video_token_id, "<|image_pad|>": cfg.image_token_id}

    def word_id(self, word: str) -> int:
        h = 2166136261
        for ch in word.encode():
            h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
        lo = min(1000, self.cfg.vocab // 2)
        hi = min(150000, self.cfg.vocab)
        return lo + h % (hi - lo)

    def __call__(self, text, add_special_tokens=False):
        import re
        out = []
        for t in text:
            ids = []
            for piece in re.split(r"(<\|vision_start\|>|<\|vision_end\|>|<\|video_pad\|>|<\|image_pad\|>)", t):
                if piece in self.special:
                    ids.append(self.special[piece])
                else:
                    ids += [self.word_id(w) for w in piece.split()]
            out.append(ids)
        return {"input_ids": out}


class SyntheticProcessor:
    """The slice of the AutoProcessor surface SGRLVRTrainer uses on video rows with the GPU front end (TR:227-229, 576):
    ``tokenizer``, ``batch_decode``, ``eos_token_id`` / ``pad_token_id``.  Decoded text is what a policy "says" for the reward
    functions: ids map to a small closed vocabulary of words that the SpaceR reward functions parse (think / answer tags,
    option letters, numbers), so ``accuracy_reward`` / ``format_reward`` do real work on every completion."""

    WORDS = ["<think>", "</think>", "<answer>", "</answer>", "A", "B", "C", "D", "the", "chair", "is", "left", "of", "table",
             "3.2", "meters", "about", "2", "objects", "behind", "sofa", "and", "door", "[", "]", ",", "1", "5", "so"]

    def __init__(self, cfg: Qwen2VLConfig):
        self.cfg = cfg
        self.eos_token_id, self.pad_token_id = cfg.eos_token_id, cfg.pad_token_id
        self.tokenizer = SyntheticTokenizer(cfg)

    def batch_decode(self, ids, skip_special_tokens=True):
        W, n = self.WORDS, len(self.WORDS)
        out = []
        for row in ids.tolist():
            toks = [t for t in row if not (skip_special_tokens and t in (self.eos_token_id, self.pad_token_id))]
            out.append(" ".join(W[t % n] for t in toks))
        return out


def synthetic_video_row(cfg: Qwen2VLConfig, prompt_idx: int, frames_u8: torch.Tensor, n_text: int, proc: SyntheticProcessor) -> dict:
    """One SpaceR-151k-shaped dataset row (SG-RLVR.py:319-352) around pre-decoded frames: a multiple-choice question whose
    rendered chat prompt tokenises to exactly ``n_text`` text tokens (template words included), so the trainer sees the
    benchmark's prompt length.  ``source_fps`` 2.0 makes the frame sampler keep every frame (QU:145-182)."""
    from .open_r1.trainer.SG_RLVR_trainer import qwen2vl_chat_template
    g = torch.Generator().manual_seed(2000 + prompt_idx)

    def row(n_words: int) -> dict:
        words = [f"w{int(v)}" for v in torch.randint(0, 100000, (n_words,), generator=torch.Generator().manual_seed(2000 + prompt_idx))]
        return dict(prompt=[{"role": "user", "content": [{"type": "video", "source_fps": 2.0}, {"type": "text", "text": " ".join(words)}]}],
                    path=frames_u8, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>",
                    problem_id=prompt_idx, options=["A. left", "B. right", "C. behind", "D. front"], data_source="synthetic")
    del g

    def n_tokens(r: dict) -> int:            # text tokens of the rendered prompt: everything but the start / pad / end markers
        return len(proc.tokenizer([qwen2vl_chat_template(r["prompt"])])["input_ids"][0]) - 3
    n = max(0, n_text - n_tokens(row(0)))
    r = row(n)
    while n_tokens(r) != n_text:             # words glued to a template marker by the renderer shift the count by one or two
        n += n_text - n_tokens(r)
        r = row(n)
    return r
