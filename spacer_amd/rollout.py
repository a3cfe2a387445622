"""Rollout engine: K sampled completions per prompt (replaces ``model.generate`` at SG_RLVR_trainer.py:463,473).

What the reference pays for K rollouts -- the ViT and the prompt prefill K times, a Python iteration and a
full-vocabulary sort per token -- is restructured for one MI355X:
  * ViT + prefill run ONCE per prompt; the prompt's K/V (post-rotary) are kept per layer and shared by the K
    rollouts of that prompt (decode attention reads [shared prompt KV | per-rollout tail KV]);
  * all prompts x K rollouts decode together as one batch of <= 64 rows, so each decode step streams the
    weights from HBM exactly once (skinny split-K GEMMs with fp32 accumulation);
  * one decode step = a fixed launch sequence whose step-dependent scalars live in device memory, captured
    once in a hipGraph and replayed (``use_graph``).
Sampling semantics: temperature -> top_k -> top_p -> multinomial (HF warper order), EOS then pad.
"""
from __future__ import annotations

import os
import weakref

from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import kernels as K
from .qwen2vl import positions as POS
from .qwen2vl.engine import Qwen2VLEngine

BF16, F32 = torch.bfloat16, torch.float32
DECODE_NORM_FOLD = os.environ.get("SPACER_DECODE_NORM", "fold") != "separate"     # A/B switch, read once at import
# decode batches of <= 16 rows (cfg4 = the reference script's 1 prompt group per GPU; cfg2): the post-attention RMSNorm is folded into the
# gate|up + SwiGLU launch (gemm_skinny_kernel SMALL + NORMA: -3 % per token-step at 8 rows of 7B, -10 % at 16 rows of 2B).
# SPACER_DECODE_SMALL=off keeps the norm launch; "attn1" (A/B) = one attention launch per layer instead of prompt split + merge
# (measured slower at 8 rows: 3.69 vs 3.28 ms per token-step -- each rollout's workgroup re-reads the prompt's keys).
_SMALL = set(filter(None, os.environ.get("SPACER_DECODE_SMALL", "fold").split(",")))
DECODE_SMALL_FOLD, DECODE_SMALL_ATTN1 = "fold" in _SMALL, "attn1" in _SMALL
# the sampler's wide form (several workgroups per row for the two passes over the logits; same tokens): SPACER_SAMPLER=narrow keeps one
# workgroup per row (A/B)
SAMPLER_WIDE = os.environ.get("SPACER_SAMPLER", "wide") != "narrow"

@dataclass
class PromptInput:
    ids: torch.Tensor                    # int64 [P] on device (unpadded prompt tokens)
    pix: Optional[torch.Tensor] = None   # bf16 [Np, patch_kpad] from spacer_patchify, or None for text-only
    grids: Optional[Sequence[Tuple[int, int, int]]] = None
    # Qwen2.5-VL: seconds per temporal grid step of each video (HF processor: temporal_patch_size / fps).  Used by the
    # ROLLOUT positions only -- the reference deletes it before its scoring forwards (SG_RLVR_trainer.py:519-520), so
    # scoring spaces the temporal positions by the default 1 s.
    second_per_grid_ts: Optional[Sequence[float]] = None
    # set by RolloutEngine.generate when it keeps the prefill's tape: the policy's scoring pass of this prompt takes its prompt-side
    # forward (ViT + prompt rows) from it instead of recomputing (Qwen2VLEngine.score_groups(prefill=...))
    prefill: Optional["PrefillSlice"] = None


@dataclass
class PrefillSlice:
    """One prompt's share of a kept prefill tape (``RolloutEngine.keep_prefill_tape``): ``shared`` = the packed pass's tapes
    ({"vit": ..., "llm": [...], "x_final", "vit_segments", "engine", "weights_version", "era_rule"}), this prompt = rows
    [row0, row0 + P) of the LLM arrays and patches [patch0, patch0 + n_patch) of the vision arrays; ``index`` = its position in the pass."""
    shared: dict
    index: int
    row0: int
    P: int
    patch0: int
    n_patch: int


@dataclass
class SamplingParams:
    max_new_tokens: int = 1024
    top_k: int = 50          # era default of transformers 4.x GenerationConfig; HF 5.x: None
    top_p: float = 0.95      # SG_RLVR_trainer.py:280
    temperature: float = 1.0  # :281
    seed: int = 0
    suppress_eos: bool = False   # fixed-length "throughput mode" of BASELINE.md
    era_rule: bool = False
    # seeded SYNTHETIC completion lengths (round 6; a random-init policy never ends a rollout by itself, a trained one does):
    # (lo, hi) -> rollout b ends with EOS as its L_b-th token, L_b ~ U[lo, hi] drawn from torch.Generator(seed) per generate call
    # (L_b > max_new_tokens: no EOS).  EOS is forbidden everywhere else (spacer_eos_schedule on the step's logits), the loop ends
    # when every row has finished, completions are padded behind EOS -- the shape of a real run's batch, with known lengths.
    synthetic_lengths: Optional[Tuple[int, int]] = None


class RolloutEngine:
    MAX_ROWS = 128  # rows per decode batch: the packed skinny GEMMs stream the weights once for up to 128 rows

    def __init__(self, engine: Qwen2VLEngine):
        self.e = engine
        self.cfg = engine.cfg
        self.dev = engine.dev
        self._packed = None          # fragment-major copies of the LLM matmul weights for the skinny GEMMs
        # decode: the input RMSNorm of every layer folded into its q|k|v projection (weights packed as W diag(w_ln1), the GEMM
        # stages the fp32 residual stream itself and sums x^2, the finishing kernel applies rstd): one launch less per layer.
        # SPACER_DECODE_NORM=separate keeps the norm kernel (A/B runs); batches of more than 64 rows always keep it.
        # (the finishing kernel of layer i clears the row sums of layer (i + 1) % layers while it reads layer i's: a one-layer model
        # would clear what it reads, so it keeps the separate norm launch)
        self.fold_norm = DECODE_NORM_FOLD and engine.cfg.layers >= 2
        self.small_fold = DECODE_SMALL_FOLD and engine.cfg.hidden % 256 == 0
        # keep the prefill's tape (ViT + prompt rows of every layer) for the policy's scoring pass: None = when it fits (the tape must
        # live through the decode loop beside the training state: ~100 GB for 8 cfg3 groups at 7B -- no; 12 GB for cfg2 at 2B -- yes),
        # True / False = forced.  Only the stored (non-recompute) policy of the Qwen2-VL tower is eligible.
        # (may also be a callable returning one of those, evaluated per generate call: GRPOEngine ties it to its live hyper-parameters)
        self.keep_prefill_tape = False
        self.prefill_tape_bytes = 0
        self.prefill_tape_prompts = 0
        # prompts per scoring pass of the caller (GRPOEngine.score_and_backward_multi takes groups_per_pass prompts): when the whole tape
        # does not fit, the auto decision keeps the tape of the FIRST passes' prompts only, in whole passes (round 6; see _tape_keep_count)
        self.prefill_pass_size = 1
        self.prefill_scored = None       # how many prompts, from the front, will be scored at all (T-GRPO twins behind them are not); None = all
        self.static_bytes = 0            # bytes that live beside the tape for the whole step (GRPOEngine: the training state); see _tape_keep_count
        self._fit_logged = set()

    # ------------------------------------------------------------------ decode-layout weights
    def invalidate(self) -> None:
        """Call after the optimizer rewrote the bf16 weights."""
        self._packed = None

    def _pack(self, rows: int = 64) -> dict:
        """Fragment-major (spacer_pack_weight_frag) copies of qkv/o/gate-up/down per layer + lm_head: every skinny-GEMM
        load instruction then reads 1 KiB of contiguous HBM.  Rebuilt once per optimizer step (~14 GB at 7B, ~10 ms).  The q|k|v and
        gate|up copies come in the form the batch's row count needs (packed on first use): q|k|v with the input-norm weight folded
        in for <= 64 rows, gate|up with the post-attention norm weight folded in for <= 16 rows."""
        W, cfg = self.e.W, self.cfg
        if self._packed is None:
            names = [f"llm.{i}.{n}" for i in range(cfg.layers) for n in ("o_w", "down_w")] + ["llm.lm_head"]
            self._packed = {n: K.pack_weight_frag(W[n]) for n in names}
        PW = self._packed
        for kind, norm, swiglu in (("gu_wn" if (self.small_fold and rows <= 16) else "gu_w", "ln2_w", True),
                                   ("qkv_wn" if (self.fold_norm and rows <= 64) else "qkv_w", "ln1_w", False)):
            if f"llm.0.{kind}" in PW:
                continue
            pack = K.pack_weight_frag_swiglu if swiglu else K.pack_weight_frag
            for i in range(cfg.layers):
                w = W[f"llm.{i}.{kind[:-1] if kind.endswith('wn') else kind}"]       # "gu_wn" -> "gu_w", "qkv_wn" -> "qkv_w"
                if kind.endswith("wn"):      # W diag(w_norm), rounded once to bf16: norm(x) W^T = rstd * (x (W diag w)^T)
                    w = (w.float() * W[f"llm.{i}.{norm}"].float()[None, :]).to(torch.bfloat16)
                PW[f"llm.{i}.{kind}"] = pack(w)
                del w
        return PW

    # ------------------------------------------------------------------ prefill
    def _tape_keep_count(self, prompts: List[PromptInput], counts: Optional[Sequence[int]] = None, C: int = 0) -> int:
        """How many of these prompts (counted from the front, in whole scoring passes) keep their prefill tape for the policy's scoring
        pass?  ``counts`` / ``C``: rollouts per prompt and tokens per rollout of this generate call (the size of the scoring passes to
        come).  Estimate: bytes per token and decoder layer of llm_forward's tape + bytes per patch and vision block of vit_forward's.
        STATIC sizes only (a decision taken from the allocator's free memory at call time made the numerics of a run depend
        on what happened to be allocated -- the reuse pass equals the full pass bit for bit only under one GEMM summation order): the
        device's memory minus what lives beside the tape for the whole step.
          * all of them when the whole tape is small: 4 x tape < memory beside the training state, tape < 15 % of the device (the scoring
            passes that follow need ~ (K C / P + 1) x the tape of their groups on top of it) -- cfg2, cfg4;
          * else (round 6) whole passes from the front while what the kept tape adds to the step's peak fits: the first pass's share
            is copied into that pass's own prompt-side tape (which the pass allocates anyway; both exist for a while: half a share,
            measured), the later passes' shares idle beside the earlier passes' activations: training state + the largest pass's own
            tape (rectangular: every rollout C tokens) + that + 7 % of the device <= the device; the kept tape under 20 % of the device.  cfg3 at 7B (8 prompts x 12.9 GB, two per pass,
            C = 512): 4 prompts = 52 GB kept, 26 GB of it idle during pass 0 -- measured HBM peak 237.5 -> 277.3 GB of 309, step 5029 ->
            4903 ms on one box; at the shipped script's C = 1024 with two groups per pass only the first pass's prompts qualify."""
        cfg = self.cfg
        per_tok = cfg.layers * (2 * 4 * cfg.hidden + 2 * (2 * cfg.hidden + cfg.qkv_dim + cfg.heads * cfg.head_dim + 3 * cfg.intermediate) + 8 + 4 * cfg.heads)
        per_patch = cfg.vit_depth * (2 * 4 * cfg.vit_dim + 2 * (6 * cfg.vit_dim + 2 * cfg.vit_mlp) + 16 + 4 * cfg.vit_heads)
        patches = [p.pix.shape[0] if p.pix is not None else 0 for p in prompts]
        per = [p.ids.numel() * per_tok + n * per_patch for p, n in zip(prompts, patches)]
        total = torch.cuda.get_device_properties(self.dev).total_memory
        avail = total - self.static_bytes
        nP, g = len(prompts), max(1, int(self.prefill_pass_size))
        n = 0
        if sum(per) * 4 < avail and sum(per) < 0.15 * total:
            n = nP
        else:
            counts = list(counts) if counts is not None else [1] * nP
            pass_dyn = max(sum(per[a:a + g]) + sum(counts[a:a + g]) * C * per_tok for a in range(0, nP, g))
            idle_budget = total - self.static_bytes - pass_dyn - 0.07 * total
            k, n_scored = g, min(nP, int(self.prefill_scored)) if self.prefill_scored else nP
            while k < nP and k <= n_scored:
                # (measured: the first pass's share is not free either -- its rows are COPIED into the pass's own tape, so for a while
                # both exist: cfg5, 1 prompt of 45.8 GB kept, peak + 21.5 GB; cfg3, 4 of 8, + 39.8 GB for 25.9 GB idle: half a share)
                need, extra = sum(per[:k]), sum(per[g:k]) + 0.5 * sum(per[:g])
                if need >= 0.20 * total or extra > idle_budget:
                    break
                n, k = k, k + g
        self.prefill_tape_bytes = sum(per[:n])
        key = (sum(p.ids.numel() for p in prompts), nP, n, C)
        if key not in self._fit_logged and len(self._fit_logged) < 8:
            self._fit_logged.add(key)
            import sys
            print(f"[spacer_amd] prefill tape: {n} of {nP} prompts keep theirs for the policy's scoring pass ({sum(per[:n]) / 1e9:.1f} of "
                  f"{sum(per) / 1e9:.1f} GB; {avail / 1e9:.0f} GB beside the training state, {g} prompt(s) per scoring pass of <= {C} tokens "
                  f"per rollout)", file=sys.stderr)
        return n

    def _prefill(self, prompts: List[PromptInput], era_rule: bool, keep_tape: bool = False, out=None):
        """ViT + LLM prefill of ALL prompts as one token-packed pass (one attention segment per prompt, one set of GEMMs
        with M = total prompt tokens): the reference runs this per rollout inside HF generate (TR:463).  ``keep_tape``: the pass
        is taped like a scoring pass and every prompt receives its ``PrefillSlice``.  ``out`` = (pk, pv, p0): write the prompt KV of
        these prompts into slots [p0, p0 + len(prompts)) of caches allocated by the caller (a taped and an untaped pass over two
        parts of the batch share one cache)."""
        cfg, e = self.cfg, self.e
        L, Hkv, D = cfg.layers, cfg.kv_heads, cfg.head_dim
        nP = len(prompts)
        plen = [p.ids.numel() for p in prompts]
        starts = [0]
        for P in plen[:-1]:
            starts.append(starts[-1] + P)
        if out is None:
            Pmax = max(plen)
            pk = torch.empty(L, nP, Pmax, Hkv, D, device=self.dev, dtype=BF16)
            pv = torch.empty_like(pk)
        else:
            pk, pv = out[0][:, out[2]:out[2] + nP], out[1][:, out[2]:out[2] + nP]        # [L, nP, Pmax, Hkv, D] views
            Pmax = pk.shape[2]
        # packed row of every (prompt, position) slot of the prompt KV cache; slots past a prompt's length are never read
        # (the attention kernels stop at plen) and take row 0
        gidx = torch.zeros(nP, Pmax, dtype=torch.int32)
        for pi, (s0, P) in enumerate(zip(starts, plen)):
            gidx[pi, :P] = torch.arange(s0, s0 + P, dtype=torch.int32)
        gidx = gidx.reshape(-1).to(self.dev)
        with_video = [p for p in prompts if p.pix is not None]
        video = None
        vit_tape, llm_tape = ({} if keep_tape else None), ([] if keep_tape else None)
        if with_video:
            pix = with_video[0].pix if len(with_video) == 1 else torch.cat([p.pix for p in with_video], 0)
            video = e.vit_forward(pix, [g for p in with_video for g in p.grids], vit_tape)
        ids = prompts[0].ids if nP == 1 else torch.cat([p.ids for p in prompts], 0)
        x0, _ = e.embed(ids, video)               # video rows fill the placeholder tokens in prompt order
        pos_list, pos_base = [], []
        for pr, P in zip(prompts, plen):
            pos3, delta = POS.mrope_positions(pr.ids.tolist(), list(pr.grids or []), cfg, era_rule, pr.second_per_grid_ts)
            pos_list.append(pos3)
            pos_base.append(P + delta)
        cos, sin = POS.mrope_tables(torch.cat(pos_list, 1), cfg, self.dev)
        segs = K.make_segments([(s0, P, 0, 0) for s0, P in zip(starts, plen)], self.dev)

        def sink(layer, k, v):          # one row gather per tensor per layer (was one device copy per prompt)
            K.gather_rows(k, gidx, out=pk[layer].view(nP * Pmax, Hkv * D))
            K.gather_rows(v, gidx, out=pv[layer].view(nP * Pmax, Hkv * D))

        x = e.llm_forward(x0, cos, sin, segs, Pmax, kv_sink=sink, tape=llm_tape)
        if keep_tape:
            shared = dict(vit=vit_tape, llm=llm_tape, x_final=x, engine=weakref.ref(e), weights_version=e.weights_version, era_rule=era_rule,
                          vit_segments=POS.vit_segments([g for p in with_video for g in p.grids]) if with_video else [])
            patch0 = 0
            for pi, (pr, s0, P) in enumerate(zip(prompts, starts, plen)):
                npatch = pr.pix.shape[0] if pr.pix is not None else 0
                pr.prefill = PrefillSlice(shared, pi, s0, P, patch0, npatch)
                patch0 += npatch
        last = torch.tensor([s0 + P - 1 for s0, P in zip(starts, plen)], dtype=torch.int64, device=self.dev)
        hn = K.rmsnorm_fwd(x.index_select(0, last), e.W["llm.norm_w"], cfg.rms_eps)
        first_logits = K.gemm_nt(hn, e.W["llm.lm_head"], out_dtype=F32)
        return pk, pv, first_logits, plen, pos_base

    # ------------------------------------------------------------------ one decode step (graph-capturable)
    def _decode_step(self, st: dict, sp: SamplingParams) -> None:
        cfg, W, PW = self.cfg, self.e.W, st["packed"]
        Hq, Hkv, D, I = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.intermediate
        B = st["B"]
        # the step's first launch also advances the device-side counters (they hold "completed steps - 1" between steps): the
        # whole step is library launches, no torch element-wise kernels
        x = K.decode_embed(st["cur_tok"], W["llm.embed"], st["x"], st["step"], st["tail_len"])
        K.decode_rope_table(st["pos_base"], st["step"], cfg.rope_theta, st["cos"], st["sin"])
        scale = D ** -0.5
        for i in range(cfg.layers):
            p = f"llm.{i}."
            if self.fold_norm and B <= 64:
                # norm(x) Wqkv^T = rstd * (bf16(x) (W diag(w))^T): the GEMM stages the fp32 stream and sums x^2 per row, the
                # finishing kernel applies rstd (and clears the next layer's row sums)
                rs = st["rowss"]
                K.gemm_skinny_packed_normed(x, PW[p + "qkv_wn"], st["acc_qkv"], rs[i], cfg.qkv_dim)
                K.decode_qkv_finish_normed(st["acc_qkv"], W[p + "qkv_b"], st["cos"], st["sin"], st["q"], st["tk"][i], st["tv"][i],
                                           st["tail_len"], rs[i], rs[(i + 1) % cfg.layers], cfg.hidden, cfg.rms_eps, Hq, Hkv, D)
            else:
                h = K.rmsnorm_fwd(x, W[p + "ln1_w"], cfg.rms_eps, out=st["h"])
                K.gemm_skinny_packed_acc(h, PW[p + "qkv_w"], st["acc_qkv"], cfg.qkv_dim)
                K.decode_qkv_finish(st["acc_qkv"], W[p + "qkv_b"], st["cos"], st["sin"], st["q"], st["tk"][i], st["tv"][i],
                                    st["tail_len"], Hq, Hkv, D)
            if st["shared_prefix"] and st["row0"] is not None:      # ... with per-prompt rollout counts (the T-GRPO twins: G / 2)
                o = K.attn_decode_shared_rows(st["q"], st["pk"][i], st["pv"][i], st["plen"], st["prompt_of"], st["row0"], st["tk"][i],
                                              st["tv"][i], st["tail_len"], st["Kn"], Hq, Hkv, D, scale, out=st["o"], workspace=st["attn_ws"])
            elif st["shared_prefix"]:      # prompt keys scored once per prompt for its K rollouts
                o = K.attn_decode_shared(st["q"], st["pk"][i], st["pv"][i], st["plen"], st["prompt_of"], st["tk"][i], st["tv"][i],
                                         st["tail_len"], st["Kn"], Hq, Hkv, D, scale, out=st["o"], workspace=st["attn_ws"])
            else:
                o = K.attn_decode(st["q"], st["pk"][i], st["pv"][i], st["plen"], st["prompt_of"], st["tk"][i], st["tv"][i],
                                  st["tail_len"], Hq, Hkv, D, scale, out=st["o"])
            K.gemm_skinny_packed_acc(o, PW[p + "o_w"], x, cfg.hidden)
            if self.small_fold and B <= 16:    # post-attention norm folded into the gate|up launch (x itself is the A operand)
                a = K.gemm_skinny_swiglu_normed(x, PW[p + "gu_wn"], I, cfg.rms_eps, out=st["a"])
            else:
                h2 = K.rmsnorm_fwd(x, W[p + "ln2_w"], cfg.rms_eps, out=st["h"])
                a = K.gemm_skinny_swiglu(h2, PW[p + "gu_w"], I, out=st["a"])      # gate|up GEMM + SwiGLU in one launch
            K.gemm_skinny_packed_acc(a, PW[p + "down_w"], x, cfg.hidden)
        hn = K.rmsnorm_fwd(x, W["llm.norm_w"], cfg.rms_eps, out=st["h"])
        if cfg.vocab >= 448 * 64:                        # whole-K workgroups: plain stores, no zero fill of the logits
            K.gemm_skinny_packed_store(hn, PW["llm.lm_head"], st["logits"], cfg.vocab)
        else:
            st["logits"].zero_()
            K.gemm_skinny_packed_acc(hn, PW["llm.lm_head"], st["logits"], cfg.vocab)
        if st.get("eos_at") is not None:                 # synthetic lengths: EOS only at the scheduled token index of each row
            K.eos_schedule_(st["logits"], st["step"], 1, st["eos_at"], cfg.eos_token_id)
        # Philox step / output column = counter + 1 = index of the token being drawn
        K.sample_top_p_step(st["logits"], st["step"], 1, st["out"], top_k=sp.top_k, top_p=sp.top_p, temperature=sp.temperature, seed=sp.seed,
                            eos_id=cfg.eos_token_id, pad_id=cfg.pad_token_id, suppress_eos=sp.suppress_eos,
                            finished=st["finished"], out_ids=st["cur_tok"], workspace=st["sample_ws"])

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def generate(self, prompts: List[PromptInput], num_generations, sp: SamplingParams, *, use_graph: bool = True,
                 stats: Optional[dict] = None, on_decode_start=None, _eos_at: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns completion ids int64 [sum of generations, C] (EOS kept, pad_token_id after it), rows ordered prompt-major like HF's
        num_return_sequences expansion.  ``num_generations``: one int for every prompt, or (round 6) one count PER PROMPT -- the T-GRPO
        twin of a sample needs only G // 2 rollouts (TR:473 ``num_return_sequences = self.shuffled_num_generations``), and a decode
        token-step costs 3.99 / 4.87 / 5.47 ms at 64 / 96 / 128 rows (scripts/probes/decode_rows_time.py): a step's main + twin
        rollouts decode as 8 x 8 + 8 x 4 = 96 rows instead of 128.  Prompt p then owns the rows [row0[p], row0[p + 1]) of the batch
        (``spacer_attn_decode_shared_rows``: the shared-prefix decode attention with per-prompt column counts; a first version ran
        non-uniform counts as virtual prompts of gcd(counts) rollouts with repeated KV slots -- twice the prompt-key workgroups for the
        main prompts, 5.43 instead of 4.9 ms per token-step).  ``on_decode_start()`` is called once, right after
        the decode step has been captured (graph capture synchronises the device, so work meant to run BESIDE the decode loop on
        another stream must be enqueued after it): the hook for prompt-only work of the step that can overlap the HBM-bound loop."""
        cfg = self.cfg
        C = sp.max_new_tokens
        counts = [int(num_generations)] * len(prompts) if isinstance(num_generations, int) else [int(k) for k in num_generations]
        assert len(counts) == len(prompts) and all(k >= 1 for k in counts)
        uniform = len(set(counts)) == 1
        Kn = max(counts)                                  # rows per prompt (uniform) / the most rows any prompt owns
        nP = len(prompts)
        B = sum(counts)
        if sp.synthetic_lengths is not None and _eos_at is None:
            lo, hi = sp.synthetic_lengths
            gen = torch.Generator().manual_seed(1_000_003 * sp.seed + 17)
            _eos_at = (torch.randint(max(1, lo), hi + 1, (B,), generator=gen) - 1).int()      # token index of EOS (length - 1), host
        if B > self.MAX_ROWS:
            if nP == 1:
                raise ValueError(f"generate: {B} rollouts of one prompt exceed the decode batch of {self.MAX_ROWS} rows")
            outs, a, r0 = [], 0, 0
            while a < nP:                                  # consecutive prompts, at most MAX_ROWS rows per decode batch
                b, rows = a, 0
                while b < nP and (b == a or rows + counts[b] <= self.MAX_ROWS):
                    rows += counts[b]
                    b += 1
                outs.append(self.generate(prompts[a:b], counts[a:b], sp, use_graph=use_graph, stats=stats,
                                          on_decode_start=on_decode_start if a == 0 else None,
                                          _eos_at=None if _eos_at is None else _eos_at[r0:r0 + rows]))
                a, r0 = b, r0 + rows
            return torch.cat(outs, 0)
        dev = self.dev
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)] if stats is not None else None
        if ev:
            ev[0].record()
        for pr in prompts:
            pr.prefill = None
        keep = self.keep_prefill_tape() if callable(self.keep_prefill_tape) else self.keep_prefill_tape
        eligible = (not self.e.recompute and cfg.vit_kind == "qwen2" and all((p.pix is None) == (prompts[0].pix is None) for p in prompts))
        n_keep = 0
        if keep is not False and eligible:
            n_auto = self._tape_keep_count(prompts, counts, C)       # (also records the kept tape's estimated size)
            n_keep = n_auto if keep is None else len(prompts)
        if n_keep == 0:
            self.prefill_tape_bytes = 0
        self.prefill_tape_prompts = n_keep               # (of len(prompts), counted from the front)
        if n_keep in (0, len(prompts)):
            pk, pv, first_logits, plen, pos_base = self._prefill(prompts, sp.era_rule, keep_tape=n_keep > 0)
        else:
            # the first n_keep prompts as a taped pass, the others as an untaped one, into one prompt-KV cache
            Pmax_all = max(p.ids.numel() for p in prompts)
            pk = torch.empty(cfg.layers, len(prompts), Pmax_all, cfg.kv_heads, cfg.head_dim, device=dev, dtype=BF16)
            pv = torch.empty_like(pk)
            _, _, fl_a, plen_a, pb_a = self._prefill(prompts[:n_keep], sp.era_rule, keep_tape=True, out=(pk, pv, 0))
            _, _, fl_b, plen_b, pb_b = self._prefill(prompts[n_keep:], sp.era_rule, keep_tape=False, out=(pk, pv, n_keep))
            first_logits, plen, pos_base = torch.cat([fl_a, fl_b], 0), plen_a + plen_b, pb_a + pb_b
        if ev:
            ev[1].record()
        L, Hkv, D, H = cfg.layers, cfg.kv_heads, cfg.head_dim, cfg.hidden
        st = dict(
            B=B, pk=pk, pv=pv, packed=self._pack(B), Kn=Kn, shared_prefix=Kn * (cfg.heads // cfg.kv_heads) <= 64 and Kn > 1 and not (DECODE_SMALL_ATTN1 and B <= 16),
            row0=None if uniform else torch.tensor([sum(counts[:i]) for i in range(nP + 1)], dtype=torch.int32, device=dev),
            attn_ws=torch.empty(K.attn_decode_workspace_bytes(nP, cfg.kv_heads) // 4, device=dev, dtype=F32),
            plen=torch.tensor(plen, dtype=torch.int32, device=dev),
            prompt_of=torch.repeat_interleave(torch.arange(nP, dtype=torch.int32), torch.tensor(counts)).to(dev),
            pos_base=torch.repeat_interleave(torch.tensor(pos_base, dtype=torch.int32), torch.tensor(counts)).to(dev).contiguous(),
            tk=K.zeros(L, B, C, Hkv, D, device=dev, dtype=BF16), tv=K.zeros(L, B, C, Hkv, D, device=dev, dtype=BF16),
            tail_len=torch.full((1,), -1, dtype=torch.int32, device=dev), step=torch.full((1,), -1, dtype=torch.int32, device=dev),
            finished=torch.zeros(B, dtype=torch.int32, device=dev), cur_tok=torch.empty(B, dtype=torch.int64, device=dev),
            x=torch.empty(B, H, device=dev, dtype=F32), h=torch.empty(B, H, device=dev, dtype=BF16),
            q=torch.empty(B, cfg.heads * D, device=dev, dtype=BF16), o=torch.empty(B, cfg.heads * D, device=dev, dtype=BF16),
            a=torch.empty(B, cfg.intermediate, device=dev, dtype=BF16),
            acc_qkv=torch.zeros(B, cfg.qkv_dim, device=dev, dtype=F32),
            rowss=torch.zeros(cfg.layers, max(B, 1), device=dev, dtype=F32),     # per layer: sum of x^2 per row (norm-folded q|k|v)
            cos=torch.empty(B, D, device=dev, dtype=F32), sin=torch.empty(B, D, device=dev, dtype=F32),
            logits=torch.empty(B, cfg.vocab, device=dev, dtype=F32),
            sample_ws=K.sample_workspace(B, cfg.vocab, dev) if SAMPLER_WIDE else None,
        )
        out = st["out"] = torch.full((B, C), cfg.pad_token_id, dtype=torch.int64, device=dev)
        st["eos_at"] = None if _eos_at is None else _eos_at.to(dev)
        if st["eos_at"] is not None and sp.suppress_eos:
            raise ValueError("SamplingParams: synthetic_lengths schedules EOS itself; suppress_eos must be False")
        # token 0 of every rollout comes from the prompt's last-position logits (K independent draws per prompt)
        st["logits"].copy_(first_logits.index_select(0, st["prompt_of"].long()))
        if st["eos_at"] is not None:
            K.eos_schedule_(st["logits"], st["step"], 1, st["eos_at"], cfg.eos_token_id)
        K.sample_top_p_step(st["logits"], st["step"], 1, out, top_k=sp.top_k, top_p=sp.top_p, temperature=sp.temperature, seed=sp.seed,
                            eos_id=cfg.eos_token_id, pad_id=cfg.pad_token_id, suppress_eos=sp.suppress_eos,
                            finished=st["finished"], out_ids=st["cur_tok"], workspace=st["sample_ws"])
        graph = None
        n_steps = 0
        for s in range(1, C):
            if use_graph and graph is None and s == 2:       # step 1 ran eagerly (warm-up); capture step 2, replay after
                graph = torch.cuda.CUDAGraph()
                # thread_local: every launch of the step comes from this thread; a device query from another thread (the RCCL
                # watchdog of a multi-rank job polling its events) must not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    self._decode_step(st, sp)
                graph.replay()                                # capture records only; this runs step 2
                if on_decode_start is not None:
                    on_decode_start()
                    on_decode_start = None
            elif graph is not None:
                graph.replay()
            else:
                self._decode_step(st, sp)
                if on_decode_start is not None and (not use_graph or s >= 2):
                    on_decode_start()
                    on_decode_start = None
            n_steps += 1
            # (every 8 token-steps: one small reduction + host read per ~30 ms of decode; at the reference's launch shape -- one group per
            # GPU -- a step's rollout ends when its 8-12 rows have ended, and a 32-step granularity left ~50 ms of a ~2 s step on the table)
            if not sp.suppress_eos and s % 8 == 0 and bool(st["finished"].all()):
                break
        if stats is not None:
            ev[2].record()
            stats["decode_steps"] = stats.get("decode_steps", 0) + n_steps
            stats["graph"] = graph is not None
            stats.setdefault("events", []).append(tuple(ev))      # (start, prefill done, decode done), HIP events
        return out
