"""The two MODEL call shapes the reference trainer relies on, kept as a thin adapter over the HIP engine:

    unwrapped_model.generate(**prompt_inputs, generation_config=self.generation_config)  -> LongTensor (K, P + C)   (TR:463)
    model(input_ids, **kwargs).logits                                                     -> (B, S, V)               (TR:357)

``SGRLVRTrainer`` itself does not go through them (it calls ``RolloutEngine.generate`` / ``Qwen2VLEngine.score_group``, which
share the prompt across the K rollouts and never materialise (B, S, V) logits); the adapter exists so that code written against
the reference's model object -- ``_get_per_token_logps`` (TR:353-366), an evaluation loop, a custom trainer -- runs unchanged
on this engine.  ``prompt_inputs`` is what the HF processor returns (TR:417-425): ``input_ids`` (1, P), ``attention_mask``,
``pixel_values_videos`` (Np, 1176) + ``video_grid_thw`` (1, 3) or ``pixel_values`` + ``image_grid_thw``, optional
``second_per_grid_ts``.  One prompt per call, left padding stripped through the attention mask, like the reference (batch 1).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch

from . import kernels as K
from .qwen2vl import positions as POS
from .qwen2vl.engine import Qwen2VLEngine
from .rollout import PromptInput, RolloutEngine, SamplingParams

F32 = torch.float32


class SpacerModel:
    """``model``-shaped view of a ``Qwen2VLEngine`` (+ its ``RolloutEngine``)."""

    def __init__(self, engine: Qwen2VLEngine, roll: Optional[RolloutEngine] = None, *, era_rule: bool = False, seed: int = 0,
                 precise: bool = False):
        self.engine, self.roll = engine, roll or RolloutEngine(engine)
        # precise = True: ``model(input_ids, ...).logits`` through the precise scoring mode (csrc/precise.hip: (hi, lo) operand pairs,
        # fp32 between operators) -- log-probs within 1e-3 of an fp32 evaluation at full depth, 2-3x the time of the fast path
        self.precise = precise
        self.cfg, self.device = engine.cfg, engine.dev
        self.era_rule, self.seed, self._calls = era_rule, seed, 0
        self.config = SimpleNamespace(vocab_size=engine.cfg.vocab, hidden_size=engine.cfg.hidden,
                                      pad_token_id=engine.cfg.pad_token_id, eos_token_id=engine.cfg.eos_token_id)

    # ------------------------------------------------------------------ helpers
    def _pixels(self, kw: dict, copies: int = 1):
        """(pix bf16 [Np, patch_kpad], grids) from processor-style kwargs; ``copies`` > 1: the reference repeats the pixel rows
        once per sequence (TR:517-518 ``pixel_values_videos.repeat(G, 1)``) -- only the first copy is used."""
        key = "pixel_values_videos" if kw.get("pixel_values_videos") is not None else ("pixel_values" if kw.get("pixel_values") is not None else None)
        if key is None:
            return None, None
        pv = kw[key]
        g = kw["video_grid_thw" if key == "pixel_values_videos" else "image_grid_thw"]
        grids = [tuple(int(v) for v in row) for row in g.tolist()]
        if copies > 1:
            assert pv.shape[0] % copies == 0 and len(grids) % copies == 0
            pv, grids = pv[:pv.shape[0] // copies], grids[:len(grids) // copies]
        pix = torch.zeros(pv.shape[0], self.cfg.patch_kpad, device=self.device, dtype=torch.bfloat16)
        pix[:, :pv.shape[1]] = pv.to(self.device).to(torch.bfloat16)
        return pix, grids

    # ------------------------------------------------------------------ TR:463
    @torch.no_grad()
    def generate(self, input_ids=None, attention_mask=None, generation_config=None, **kw) -> torch.Tensor:
        """K = generation_config.num_return_sequences sampled continuations of ONE prompt; returns (K, P + C) ids: the prompt
        (as given, pads included) followed by the completion, EOS kept and pad_token_id after it -- HF's layout, so
        ``prompt_completion_ids[:, P:]`` is the completion (TR:465-466)."""
        gc = generation_config
        assert input_ids is not None and input_ids.shape[0] == 1, "one prompt per generate call, as in the reference"
        ids = input_ids[0]
        if attention_mask is not None:
            ids = ids[attention_mask[0].bool()]                      # strip the left padding
        pix, grids = self._pixels(kw)
        sec = kw.get("second_per_grid_ts")
        if sec is not None:
            sec = [float(v) for v in (sec.tolist() if hasattr(sec, "tolist") else sec)]
        n = int(getattr(gc, "num_return_sequences", 1) or 1)
        do_sample = bool(getattr(gc, "do_sample", False))
        top_k = getattr(gc, "top_k", None)
        sp = SamplingParams(max_new_tokens=int(gc.max_new_tokens), top_k=(int(top_k) if top_k else (0 if do_sample else 1)) if do_sample else 1,
                            top_p=float(getattr(gc, "top_p", 1.0) or 1.0) if do_sample else 1.0,
                            temperature=float(getattr(gc, "temperature", 1.0) or 1.0), seed=self.seed + 7919 * self._calls,
                            era_rule=self.era_rule)
        if sp.top_k == 0:
            sp.top_k = self.cfg.vocab                                # HF top_k=None: no top-k cut
        self._calls += 1
        comp = self.roll.generate([PromptInput(ids.to(self.device).long(), pix, grids, sec)], n, sp)
        return torch.cat([input_ids.to(self.device).long().expand(n, -1), comp], dim=1)

    # ------------------------------------------------------------------ TR:357
    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, **kw):
        """Teacher-forced forward of B full sequences -> object with ``.logits`` fp32 (B, S, V): the tensor the reference's
        ``_get_per_token_logps`` slices.  Each row is an independent causal sequence (no padding mask, TR:357 passes none);
        the pixel rows may be given once or repeated B times (TR:517-518)."""
        e, cfg = self.engine, self.cfg
        B, S = input_ids.shape
        pix, grids = self._pixels(kw, copies=B if (kw.get("video_grid_thw") is not None and kw["video_grid_thw"].shape[0] == B and B > 1)
                                  or (kw.get("image_grid_thw") is not None and kw["image_grid_thw"].shape[0] == B and B > 1) else 1)
        unit_rev = None
        if pix is None:
            video = None
        elif self.precise:
            video, unit_rev = e._vit_forward_precise(pix, grids)
        else:
            video = e.vit_forward(pix, grids)
        out = torch.empty(B, S, cfg.vocab, device=self.device, dtype=F32)
        for b in range(B):
            ids = input_ids[b].to(self.device).long()
            # only the first Nv placeholder ids are placeholders (they sit in the prompt); a SAMPLED <|video_pad|> / <|image_pad|>
            # further on is an ordinary token with its own embedding row and a text position, as in Qwen2VLEngine.score_groups
            cut = S
            if video is not None:
                is_vis = (ids == cfg.video_token_id) | (ids == cfg.image_token_id)
                nth = torch.nonzero(is_vis).reshape(-1)
                assert nth.numel() >= video.shape[0], "fewer placeholder tokens than vision rows"
                cut = int(nth[video.shape[0] - 1]) + 1
            x0, _ = e.embed(ids, video, placeholder_scopes=[(0, cut)] if video is not None else None, unit_rev=unit_rev)
            pos3, delta = POS.mrope_positions(ids[:cut].tolist(), list(grids or []), cfg, self.era_rule)
            if cut < S:
                tail = (cut + delta) + torch.arange(S - cut)
                pos3 = torch.cat([pos3, tail.view(1, -1).expand(3, -1)], dim=1)
            cos, sin = POS.mrope_tables(pos3, cfg, self.device)
            segs = K.make_segments([(0, S, 0, 0)], self.device)
            if self.precise:
                x = e._llm_forward_precise(x0, cos, sin, segs, S)
                K.gemm_pair(*K.norm_pair(x, e.W["llm.norm_w"], None, cfg.rms_eps), e.W["llm.lm_head"], out=out[b])
                continue
            x = e.llm_forward(x0, cos, sin, segs, S)
            hn = K.rmsnorm_fwd(x, e.W["llm.norm_w"], cfg.rms_eps)
            K.gemm_nt(hn, e.W["llm.lm_head"], out=out[b], out_dtype=F32)
        return SimpleNamespace(logits=out)

    def eval(self):
        return self

    def train(self, mode: bool = True):
        return self
