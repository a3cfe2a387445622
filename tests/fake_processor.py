"""A stand-in for AutoProcessor with the call surface the trainer uses (TR:417-425, 576): whitespace "tokenizer",
placeholder expansion and HF-layout patchify (via the oracle's restatement).  Test infrastructure only."""
import torch

from oracle import qwen2vl_fp32 as O


class FakeTokenizer:
    """``processor.tokenizer`` surface: text with special markers -> ids, one id per <|video_pad|> / <|image_pad|>."""

    def __init__(self, proc):
        self.p = proc

    def __call__(self, text, add_special_tokens=False):
        import re
        cfg = self.p.cfg
        special = {"<|vision_start|>": cfg.vision_start_id, "<|vision_end|>": cfg.vision_end_id,
                   "<|video_pad|>": cfg.video_token_id, "<|image_pad|>": cfg.image_token_id}
        out = []
        for t in text:
            ids = []
            for piece in re.split(r"(<\|vision_start\|>|<\|vision_end\|>|<\|video_pad\|>|<\|image_pad\|>)", t):
                if piece in special:
                    ids.append(special[piece])
                else:
                    ids += [self.p._tok(w) for w in piece.split()]
            out.append(ids)
        return {"input_ids": out}


class FakeProcessor:
    def __init__(self, cfg, with_tokenizer: bool = True):
        self.cfg = cfg
        self.eos_token_id = cfg.eos_token_id
        self.pad_token_id = cfg.pad_token_id
        self.ocfg = cfg.as_oracle_dict()
        if with_tokenizer:
            self.tokenizer = FakeTokenizer(self)

    def _tok(self, word):
        return 10 + (sum(ord(c) * (i + 1) for i, c in enumerate(word)) % 900)

    def __call__(self, text, images=None, videos=None, return_tensors="pt", padding=True, padding_side="left",
                 add_special_tokens=False):
        assert len(text) == 1
        marker = "<|vision_start|><|video_pad|><|vision_end|>"
        imarker = "<|vision_start|><|image_pad|><|vision_end|>"
        out = {}
        ids = []
        if imarker in text[0]:            # an image row: one still frame, duplicated over the temporal patch like HF's image processor
            import numpy as np
            head, tail = text[0].split(imarker)
            frame = torch.from_numpy(np.asarray(images[0]).copy()).permute(2, 0, 1).unsqueeze(0)
            rows, grid = O.patchify_frames(frame, self.ocfg)
            nv = grid[0] * grid[1] * grid[2] // (self.cfg.merge ** 2)
            ids = [self._tok(w) for w in head.split()] + [self.cfg.vision_start_id] + [self.cfg.image_token_id] * nv \
                + [self.cfg.vision_end_id] + [self._tok(w) for w in tail.split()]
            out["pixel_values"], out["image_grid_thw"] = rows, torch.tensor([list(grid)])
            out["input_ids"] = torch.tensor([ids])
            out["attention_mask"] = torch.ones_like(out["input_ids"])
            return out
        parts = text[0].split(marker)
        for i, part in enumerate(parts):
            ids += [self._tok(w) for w in part.split()]
            if i + 1 < len(parts):
                frames = videos[0].round().clamp(0, 255).to(torch.uint8)
                rows, grid = O.patchify_frames(frames, self.ocfg)
                nv = grid[0] * grid[1] * grid[2] // (self.cfg.merge ** 2)
                ids += [self.cfg.vision_start_id] + [self.cfg.video_token_id] * nv + [self.cfg.vision_end_id]
                out["pixel_values_videos"] = rows
                out["video_grid_thw"] = torch.tensor([list(grid)])
        out["input_ids"] = torch.tensor([ids])
        out["attention_mask"] = torch.ones_like(out["input_ids"])
        return out

    def batch_decode(self, ids, skip_special_tokens=True):
        res = []
        for row in ids.tolist():
            toks = [t for t in row if not (skip_special_tokens and t in (self.eos_token_id, self.pad_token_id))]
            res.append(" ".join(f"w{t}" for t in toks))
        return res
