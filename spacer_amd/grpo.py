"""SG-RLVR / GRPO step on the HIP engine: rollout -> reward shaping -> policy/ref scoring -> loss -> backward ->
data-parallel gradient reduction -> AdamW.  This is the arithmetic of ``SGRLVRTrainer.compute_loss``
(SG_RLVR_trainer.py:384-686) plus the optimizer step HF Trainer / DeepSpeed run after it, with the reference's
redundancy removed (ViT once per weights, prompt prefilled once, logits only on completion rows).

Host-side reward shaping (T-GRPO bonus :598-617, length bonus :620-629, group advantage :632-638) stays
plain Python/torch on K-element tensors, as in the reference; everything token-sized runs in libspacer_hip.so.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import kernels as K
from .qwen2vl.config import Qwen2VLConfig
from .qwen2vl.engine import Qwen2VLEngine
from .qwen2vl.weights import FlatParams
from .rollout import PromptInput, RolloutEngine, SamplingParams

F32 = torch.float32


@dataclass
class GRPOHyper:
    num_generations: int = 8          # --num_generations 8   (run_SpaceR_SG_RLVR.sh:39)
    beta: float = 0.04                # --beta 0.04           (:36)
    learning_rate: float = 1e-6       # --learning_rate 1e-6  (:22)
    weight_decay: float = 0.01        # --weight_decay 0.01   (:24)
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_eps: float = 1e-8
    max_grad_norm: float = 5.0        # --max_grad_norm 5     (:37)
    temporal: bool = True             # --temporal true       (:19)
    len_control: bool = True          # --len_control true    (:20)
    lr_scheduler_type: str = "cosine"  # (:23)
    total_steps: int = 1000
    warmup_steps: int = 0
    # gradient all-reduce on a bf16 wire: the communication dtype of the reference's DeepSpeed bf16 configuration
    # (local_scripts/zero3.json:14-33 "bf16": {"enabled": "auto"}); fp32 values on the wire when False
    grad_comm_bf16: bool = True
    overlap_comm: bool = True         # zero3.json:26 "overlap_comm": true -- reduce layer ranges while the backward still runs
    # --gradient_checkpointing true (run_SpaceR_SG_RLVR.sh:28): selective activation recompute in the policy's backward
    # (Qwen2VLEngine.recompute: MLP intermediates + lm_head logits; bit-identical gradients, -20 GB of saved activations per 7B
    # prompt group for one more gate|up and lm_head GEMM).  Off by default: 288 GB holds two groups per pass without it.
    recompute: bool = False
    # the rollout's prefill keeps its tape and the policy's scoring pass takes the prompt-side forward (ViT + prompt rows) from it
    # instead of recomputing it (the reference runs it twice: TR:463 inside generate, TR:517-541 in the scoring forward).  None = when
    # the tape fits beside the training state (2B: yes; 7B with 8 groups per GPU: ~100 GB, no), True / False = forced
    reuse_prefill: Optional[bool] = None
    # data-parallel gradient exchange: "allreduce" = bucketed all-reduce of the flat gradient, overlapped with the last backward, then
    # the SAME AdamW on every replica; "rs_ag" = reduce-scatter of the gradient, AdamW on the rank's 1/world shard of the
    # parameters only, all-gather of the updated bf16 weights (on the xGMI mesh every GPU talks to its 7 peers
    # directly, so the reduce-scatter + all-gather pair moves (n-1)/n x 2 bytes per parameter in bf16 instead of the all-reduce's
    # (n-1)/n x 4, and the optimizer's 16 B/param of HBM traffic shrinks by world x).  Replicas end bit-identical either way.
    grad_algo: str = "allreduce"
    # --precise_logps: the per-token log-probs of the policy AND the reference model that enter KL, loss and the logged metrics
    # (TR:527-552) are evaluated in the precise mode (csrc/precise.hip: (hi, lo) bf16 operand pairs, two-pass GEMMs, pair
    # attention, fp32 everywhere else) -- within the north-star's 1e-3 of an fp32 evaluation at full 7B depth, where the fast path
    # sits at the bf16-operand floor (~1e-1 max).  The precise forward of the policy also emits the tape of the production
    # backward (Qwen2VLEngine._llm_forward_precise), so the gradient costs nothing extra; the forward costs ~2x the fast one.
    precise_logps: bool = False
    # EOS-trimmed scoring (round 6): the reference + policy scoring passes and the backward pack only the tokens up to each rollout's
    # first EOS.  TR:493-498 builds the completion mask and TR:640-643 multiplies it into the loss, so everything behind the first
    # EOS contributes exactly zero; with --max_completion_length 1024 and a length bonus that pays for 320-512 tokens (TR:620-629)
    # about half of the [K, C] rectangle of a real run is pad.  Same log-probs on the unmasked positions, same loss / KL / gradient
    # (tests/test_ragged_gpu.py); costs one host read of K lengths per pass.  False = the rectangular pass (A/B, tests).
    trim_completions: bool = True


# ------------------------------------------------------------------------------------- reward shaping (host)
def temporal_bonus(rewards_per_func: torch.Tensor, shuffled_rewards_per_func: Optional[torch.Tensor], temporal: bool,
                   has_video: bool) -> Tuple[torch.Tensor, float]:
    """SG_RLVR_trainer.py:598-617.  Column 0 = accuracy reward.  Returns (summed rewards [G], temporal_reward)."""
    if temporal and has_video:
        t = rewards_per_func.clone()
        # (fp32 tensor arithmetic as the reference's line: 0.8 * mean in float32, not in Python doubles -- ties decide the bonus)
        if bool(t[:, 0].mean() >= 0.8 * shuffled_rewards_per_func[:, 0].mean()):
            sel = t[:, 0] > 0.1
            t[sel, 0] += 0.3
            return t.sum(dim=1), 1.0
        return t.sum(dim=1), 0.0
    return rewards_per_func.sum(dim=1), 0.5


def length_bonus(rewards: torch.Tensor, rewards_per_func: torch.Tensor, lengths: torch.Tensor, len_control: bool) -> torch.Tensor:
    """SG_RLVR_trainer.py:620-629: +0.2 for 320 <= len <= 512, only when more than one rollout has acc > 0.1."""
    out = rewards.clone()
    if len_control:
        # (tensor comparison as TR:622: an accuracy reward of exactly float32(0.1) is NOT > 0.1; ``float(x) > 0.1`` in Python doubles
        # says it is -- found by the golden table of the reference's own lines, tests/golden/grpo_lines.json case "len3")
        idx = torch.nonzero(rewards_per_func[:, 0] > 0.1, as_tuple=True)[0].tolist()
        if len(idx) > 1:
            for i in idx:
                if 320 <= int(lengths[i]) <= 512:
                    out[i] += 0.2
    return out


def group_advantages(rewards: torch.Tensor, num_generations: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """SG_RLVR_trainer.py:632-638: (r - mean_g) / (std_g + 1e-4) with torch's unbiased std."""
    g = rewards.view(-1, num_generations)
    mean = g.mean(dim=1, keepdim=True)
    std = g.std(dim=1, keepdim=True)
    return ((g - mean) / (std + 1e-4)).reshape(-1), std.expand_as(g).reshape(-1)


def lr_at(step: int, h: GRPOHyper) -> float:
    """HF get_scheduler semantics for "cosine" / "linear" / "constant" with optional warm-up; step counts from 0."""
    if step < h.warmup_steps:
        return h.learning_rate * (step + 1) / max(1, h.warmup_steps)
    if h.lr_scheduler_type == "constant":
        return h.learning_rate
    prog = (step - h.warmup_steps) / max(1, h.total_steps - h.warmup_steps)
    prog = min(max(prog, 0.0), 1.0)
    if h.lr_scheduler_type == "linear":
        return h.learning_rate * (1.0 - prog)
    return h.learning_rate * 0.5 * (1.0 + math.cos(math.pi * prog))


def allreduce_flat_(flat: torch.Tensor, pg, bucket_elems: int = 1 << 28, wire_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """SUM all-reduce of one flat tensor in a few large buckets (1 GiB of fp32 each by default), all in flight at
    once.  On MI355X the 8 GPUs are fully connected by point-to-point xGMI links, so collectives are per-link bound:
    few, big messages; the mean (1/world) is folded into the optimizer's grad_scale instead of a second pass.

    wire_dtype = torch.bfloat16 halves the bytes on the links: each bucket is rounded to bf16, summed in bf16 by the
    collective and widened back into ``flat`` -- what the reference's DeepSpeed bf16 configuration does with its gradient
    buckets (communication in the training dtype).  Default: exchange the fp32 values themselves."""
    import torch.distributed as dist
    if wire_dtype is None or wire_dtype == flat.dtype:
        works = [dist.all_reduce(flat[a:a + bucket_elems], group=pg, async_op=True) for a in range(0, flat.numel(), bucket_elems)]
        for w in works:
            w.wait()
        return flat
    pending = []
    for a in range(0, flat.numel(), bucket_elems):
        wire = flat[a:a + bucket_elems].to(wire_dtype)
        pending.append((a, wire, dist.all_reduce(wire, group=pg, async_op=True)))
    for a, wire, w in pending:
        w.wait()
        flat[a:a + wire.numel()].copy_(wire)
    return flat


class GradReducer:
    """Bucketed data-parallel gradient exchange OVERLAPPED with the backward pass (the reference's contract:
    local_scripts/zero3.json:26 ``overlap_comm``; DeepSpeed reduces gradient buckets as autograd produces them).

    The flat gradient buffer is in layer order and the backward of the rank's LAST prompt group finalises it from the
    end (lm_head, decoder layers 27..0, embedding, merger, vision blocks 31..0, patch embedding).  The engine reports each
    finished parameter range (``ready``); ranges are merged into buckets of >= ``bucket_elems`` and every full bucket is
    handed to RCCL at once on a side stream: the collective of layer i runs under the GEMMs of layers i-1, i-2, ...  On
    MI355X the 8 GPUs are joined by point-to-point xGMI links, so a collective is per-link bound -- buckets are big
    (>= 128 Mi elements: a 7B decoder layer, 233 M parameters, goes out as ONE message the moment its backward ends) and few.  ``finish`` drains what is left and makes the compute stream
    wait for the side stream.  wire_dtype = bf16 halves the bytes: a bucket is rounded to bf16 on the compute stream, summed
    by the collective and widened back into the fp32 buffer."""

    def __init__(self, flat: torch.Tensor, specs, pg, *, wire_dtype: Optional[torch.dtype] = None, bucket_elems: int = 1 << 27):
        self.flat, self.pg, self.wire_dtype, self.bucket_elems = flat, pg, wire_dtype, bucket_elems
        self.ranges = {}
        starts = [s.offset for s in specs] + [flat.numel()]
        end = {s.name: starts[i + 1] for i, s in enumerate(specs)}       # up to the next tensor: alignment padding included
        for s in specs:                                     # parameter-name prefix -> [lo, hi)
            for pre in {s.name, s.name.rsplit(".", 1)[0] + "." if "." in s.name else s.name}:
                lo, hi = self.ranges.get(pre, (s.offset, end[s.name]))
                self.ranges[pre] = (min(lo, s.offset), max(hi, end[s.name]))
        import torch.distributed as dist
        # gloo moves host memory: device gradients are staged through the host (tests of the multi-rank control flow on a
        # one-GPU box; RCCL ("nccl") takes the device buffers directly on a side stream)
        self.host_staged = flat.is_cuda and dist.get_backend(pg) == "gloo"
        self.cuda = flat.is_cuda and not self.host_staged
        self.side = torch.cuda.Stream(device=flat.device) if self.cuda else None
        # bench.py's ``comm`` object: (event at finish() entry, event when the compute stream may go on) per step = the part of the
        # exchange NOT hidden under the last backward; bytes handed to the collective per step
        self.timing = False
        self.exposed = []
        self.wire_bytes = 0
        self.reset()

    def reset(self) -> None:
        self.pending: List[Tuple[int, int]] = []          # finished, not yet sent
        self.inflight = []                                  # (lo, hi, wire or None, work)
        self.sent = 0

    def _send(self, lo: int, hi: int) -> None:
        import torch.distributed as dist
        view = self.flat[lo:hi]
        if self.host_staged:
            wire = view.to("cpu", self.wire_dtype or view.dtype)
        else:
            wire = view.to(self.wire_dtype) if self.wire_dtype not in (None, view.dtype) else None
        buf = wire if wire is not None else view
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.side.wait_event(ev)
                work = dist.all_reduce(buf, group=self.pg, async_op=True)
        else:
            work = dist.all_reduce(buf, group=self.pg, async_op=True)
        self.inflight.append((lo, hi, wire, work))
        self.sent += hi - lo
        self.wire_bytes += (hi - lo) * buf.element_size()

    def _flush(self, force: bool) -> None:
        """Merge adjacent finished ranges; send every merged run that reached the bucket size (all of them when forced)."""
        self.pending.sort()
        merged: List[List[int]] = []
        for lo, hi in self.pending:
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        self.pending = []
        for lo, hi in merged:
            if force or hi - lo >= self.bucket_elems:
                for a in range(lo, hi, 4 * self.bucket_elems):
                    self._send(a, min(hi, a + 4 * self.bucket_elems))
            else:
                self.pending.append((lo, hi))

    def ready(self, prefix: str) -> None:
        """The gradient of every parameter whose name starts with ``prefix`` (a layer prefix such as "llm.12." or a full
        name) is final on the current stream."""
        self.pending.append(self.ranges[prefix])
        self._flush(False)

    def finish(self) -> None:
        """Send what has not been sent (everything, if no ``ready`` call was made), wait, widen bf16 wires back."""
        t_in = None
        if self.timing and self.cuda:
            t_in = torch.cuda.Event(enable_timing=True)
            t_in.record(torch.cuda.current_stream())
        covered = sorted([(lo, hi) for lo, hi, _, _ in self.inflight] + self.pending)
        pos, gaps = 0, []
        for lo, hi in covered:
            if lo > pos:
                gaps.append((pos, lo))
            pos = max(pos, hi)
        if pos < self.flat.numel():
            gaps.append((pos, self.flat.numel()))
        self.pending += gaps
        self._flush(True)
        for lo, hi, wire, work in self.inflight:
            if self.cuda:
                with torch.cuda.stream(self.side):
                    work.wait()
                    if wire is not None:
                        self.flat[lo:hi].copy_(wire)
            else:
                work.wait()
                if wire is not None:
                    self.flat[lo:hi].copy_(wire)
        if self.cuda:
            torch.cuda.current_stream().wait_stream(self.side)
        if t_in is not None:
            t_out = torch.cuda.Event(enable_timing=True)
            t_out.record(torch.cuda.current_stream())
            self.exposed.append((t_in, t_out))
        assert self.sent == self.flat.numel(), (self.sent, self.flat.numel())
        self.reset()


class ShardedExchange:
    """``grad_algo = "rs_ag"``: reduce-scatter the flat gradient (rank r keeps the SUM of shard r), and after the sharded AdamW
    all-gather the updated bf16 weights.  Shard r = elements [r S, (r + 1) S) with S = ceil(n / world) rounded up to 64 elements
    (the last shard is short).  The gradient travels in buckets of ``bucket`` elements per rank: one contiguous [world, bucket]
    wire buffer (bf16 by default) per collective, so the xGMI links carry world x bucket x 2 bytes per call.  gloo (CPU tests) has
    no reduce-scatter: there the same sums are formed by one ``reduce`` per destination rank."""

    def __init__(self, n: int, pg, *, wire_dtype: Optional[torch.dtype], bucket: int = 1 << 26):
        import torch.distributed as dist
        self.pg, self.wire_dtype = pg, wire_dtype
        self.world, self.rank = dist.get_world_size(pg), dist.get_rank(pg)
        self.n = n
        self.S = ((n + self.world - 1) // self.world + 63) // 64 * 64
        self.bucket = min(bucket, self.S)
        self.lo, self.hi = min(n, self.rank * self.S), min(n, (self.rank + 1) * self.S)
        self.native = dist.get_backend(pg) == "nccl"
        self._bufs = {}          # (kind, dtype, device) -> wire buffers, allocated ONCE (world x bucket elements: ~1 GB at world 8)

    def shard(self, flat: torch.Tensor) -> torch.Tensor:
        return flat[self.lo:self.hi]

    def drop_buffers(self, dtype) -> None:
        """Release the wire buffers of one dtype (the fp32 ones exist only around a full-state checkpoint)."""
        for key in [k for k in self._bufs if k[1] == dtype]:
            del self._bufs[key]

    def _buf(self, kind: str, shape, dtype, dev) -> torch.Tensor:
        key = (kind, dtype, str(dev))
        b = self._bufs.get(key)
        if b is None or tuple(b.shape) != tuple(shape):
            b = self._bufs[key] = torch.empty(*shape, device=dev, dtype=dtype)
        return b

    def reduce_scatter_(self, flat: torch.Tensor) -> None:
        """flat[lo:hi] <- sum over ranks of flat[lo:hi] (the other shards of ``flat`` are left as they are: stale).  Runs after
        the last backward: ``overlap_comm`` has no effect under rs_ag (the shards interleave every layer's range)."""
        import torch.distributed as dist
        W, S, n = self.world, self.S, self.n
        wd = self.wire_dtype or flat.dtype
        dev = flat.device if self.native else torch.device("cpu")
        wire_full = self._buf("rs_wire", (W, self.bucket), wd, dev)
        out_full = self._buf("rs_out", (self.bucket,), wd, dev)
        for b0 in range(0, S, self.bucket):
            bl = min(self.bucket, S - b0)
            wire = wire_full if bl == self.bucket else wire_full.view(-1)[:W * bl].view(W, bl)
            for r in range(W):                                  # piece [b0, b0 + bl) of every rank's shard, clipped at n
                a, e = min(n, r * S + b0), min(n, r * S + b0 + bl)
                if e > a:
                    wire[r, :e - a].copy_(flat[a:e])
                if e - a < bl:                                  # only the clipped tail of the last shard needs zeros
                    wire[r, max(0, e - a):].zero_()
            if self.native:
                out = out_full[:bl]
                dist.reduce_scatter_tensor(out, wire.view(-1), group=self.pg)
            else:
                works = [dist.reduce(wire[r], dst=dist.get_global_rank(self.pg, r), group=self.pg, async_op=True) for r in range(W)]
                for w in works:
                    w.wait()
                out = wire[self.rank]
            a, e = min(n, self.rank * S + b0), min(n, self.rank * S + b0 + bl)
            if e > a:
                flat[a:e].copy_(out[:e - a])

    def all_gather_(self, flat: torch.Tensor) -> None:
        """Every rank's [lo, hi) slice of ``flat`` (bf16 weights, or any state tensor) -> all ranks."""
        import torch.distributed as dist
        W, S, n = self.world, self.S, self.n
        dev = flat.device if self.native else torch.device("cpu")
        mine_full = self._buf("ag_mine", (self.bucket,), flat.dtype, dev)
        full_full = self._buf("ag_full", (W, self.bucket), flat.dtype, dev)
        for b0 in range(0, S, self.bucket):
            bl = min(self.bucket, S - b0)
            mine = mine_full[:bl]
            a, e = min(n, self.rank * S + b0), min(n, self.rank * S + b0 + bl)
            if e > a:
                mine[:e - a].copy_(flat[a:e])
            if e - a < bl:
                mine[max(0, e - a):].zero_()
            full = full_full if bl == self.bucket else full_full.view(-1)[:W * bl].view(W, bl)
            if self.native:
                dist.all_gather_into_tensor(full.view(-1), mine, group=self.pg)
            else:
                parts = [torch.empty_like(mine) for _ in range(W)]
                dist.all_gather(parts, mine, group=self.pg)
                full = torch.stack(parts)
            for r in range(W):
                a, e = min(n, r * S + b0), min(n, r * S + b0 + bl)
                if e > a and r != self.rank:
                    flat[a:e].copy_(full[r, :e - a])


# ------------------------------------------------------------------------------------- the step engine
class GRPOEngine:
    """Owns policy + frozen reference weights, fp32 master / Adam state / gradients, and runs the step phases."""

    def __init__(self, cfg: Qwen2VLConfig, policy: FlatParams, hyper: GRPOHyper, *, ref: Optional[FlatParams] = None,
                 process_group=None):
        self.cfg, self.h = cfg, hyper
        self.dev = policy.flat.device
        self.policy = policy
        self.ref = ref if ref is not None else FlatParams(cfg, policy.flat.clone(), policy.specs)   # create_reference_model
        self.engine = Qwen2VLEngine(cfg, self.policy, recompute=hyper.recompute)
        self.ref_engine = Qwen2VLEngine(cfg, self.ref)
        self.roll = RolloutEngine(self.engine)
        # (the closure holds the hyper-parameter object, not this engine: a GRPOEngine <-> RolloutEngine reference cycle would keep
        # 165 GB of training state alive until the cyclic collector runs)
        self.roll.keep_prefill_tape = lambda h=hyper: False if (h.recompute or h.precise_logps) else h.reuse_prefill
        # the auto decision (reuse_prefill = None) is made from STATIC sizes -- the device's memory, the training state allocated below
        # (2 + 2 + 4 + 4 + 4 + 4 bytes per parameter) and the decode-layout weight copies -- not from the allocator's free memory at the
        # moment of the call: the same configuration decides the same way on every run, rank and step
        self.roll.static_bytes = policy.flat.numel() * (2 + 2 + 4 + 4 + 4 + 4 + 2)
        self.master = FlatParams(cfg, policy.flat.float(), policy.specs)
        self.G = policy.like(F32)
        self.m = torch.zeros_like(self.master.flat)
        self.v = torch.zeros_like(self.master.flat)
        self.step_count = 0
        self.pg = process_group
        self._sumsq = torch.zeros(1, device=self.dev, dtype=F32)
        self.reducer, self.sharded = None, None
        if hyper.grad_algo not in ("allreduce", "rs_ag"):
            raise ValueError(f"grad_algo {hyper.grad_algo!r}: allreduce or rs_ag")
        if process_group is not None:
            wire = torch.bfloat16 if hyper.grad_comm_bf16 else None
            if hyper.grad_algo == "rs_ag":
                self.sharded = ShardedExchange(self.G.flat.numel(), process_group, wire_dtype=wire)
            else:
                self.reducer = GradReducer(self.G.flat, self.G.specs, process_group, wire_dtype=wire)

    # -------------------------------------------------------------- phases
    def rollout(self, prompts: List[PromptInput], sp: SamplingParams, stats: Optional[dict] = None) -> torch.Tensor:
        return self.roll.generate(prompts, self.h.num_generations, sp, stats=stats)

    @torch.no_grad()
    def logps(self, prompts: List[PromptInput], completions: List[torch.Tensor], *, which: str = "policy", precise: bool = False,
              era_rule: bool = False) -> torch.Tensor:
        """Per-token log-probs [G*K, C] of the completions under the policy or the frozen reference model (TR:353-366, no
        gradient): evaluation / reporting.  ``precise=True`` = the mode that holds 1e-3 against an fp32 evaluation at full depth."""
        eng = {"policy": self.engine, "ref": self.ref_engine}[which]
        return eng.score_groups([(p.ids, p.pix, p.grids) for p in prompts], completions, era_rule=era_rule, precise=precise)

    def _take_prefill(self, prompts: List[PromptInput], era_rule: bool):
        """The prompts' shares of a kept prefill tape, detached from the prompts; None -- and the shares released BEFORE the scoring
        forwards run -- when the policy's pass cannot use them (precise / recompute mode, other weights, mixed passes)."""
        pf = [getattr(p, "prefill", None) for p in prompts]
        for p in prompts:
            p.prefill = None
        if any(x is None for x in pf) or self.h.precise_logps:
            return None
        return pf if self.engine._prefill_usable(pf, [(p.ids, p.pix, p.grids) for p in prompts], era_rule) else None

    def score_and_backward(self, prompt: PromptInput, completion_ids: torch.Tensor, advantages: torch.Tensor,
                           grad_scale: float = 1.0, *, era_rule: bool = False, last_group: bool = False) -> Dict[str, torch.Tensor]:
        """One prompt group: masks, reference + policy log-probs, loss, backward into self.G.  ``advantages`` fp32 [K]
        (device).  grad_scale folds 1/num_groups (gradient accumulation mean).  ``last_group`` = this is the rank's last
        backward before the optimizer step: finished layer ranges go to the data-parallel reducer while it runs."""
        cfg = self.cfg
        mask, lengths = K.completion_mask(completion_ids, cfg.eos_token_id)
        lens = lengths.tolist() if self.h.trim_completions else None      # (host read: the rollouts are finished by now)
        pf = self._take_prefill([prompt], era_rule)
        with torch.no_grad():
            pr = self.h.precise_logps
            ref_lp = self.ref_engine.score_group(prompt.ids, completion_ids, prompt.pix, prompt.grids, era_rule=era_rule, precise=pr, lengths=lens)
            tape: dict = {}
            lp = self.engine.score_groups([(prompt.ids, prompt.pix, prompt.grids)], [completion_ids], tape=tape, era_rule=era_rule, precise=pr,
                                          prefill=pf, lengths=lens)
            del pf                                     # (its share of the kept tape is consumed)
            if callable(advantages):           # lazy: the host shapes the rewards while the two forwards run (see _multi)
                advantages = advantages()
                advantages = (advantages[0] if isinstance(advantages, (list, tuple)) else advantages).to(self.dev)
            loss, kl, dlogp = K.grpo_loss(lp, ref_lp, advantages, mask, self.h.beta)
            if grad_scale != 1.0:
                dlogp.mul_(grad_scale)
            hook = self.reducer.ready if (last_group and self.reducer is not None and self.h.overlap_comm) else None
            self.engine.backward_group(tape, dlogp, self.G, on_ready=hook)
        return dict(loss=loss, kl=kl, logps=lp, ref_logps=ref_lp, mask=mask, lengths=lengths, scored_tokens=int(sum(lens)) if lens is not None else mask.numel())

    def score_and_backward_multi(self, prompts: List[PromptInput], completions: List[torch.Tensor], advantages: List[torch.Tensor],
                                 grad_scale: float = 1.0, *, era_rule: bool = False, last_group: bool = False) -> Dict[str, torch.Tensor]:
        """``score_and_backward`` for several prompt groups in ONE scoring pass each for the reference and the policy and ONE
        backward (Qwen2VLEngine.score_groups): the same gradients as calling it group by group WITH THE SAME ``grad_scale`` --
        ``grad_scale`` is the weight of EACH group in the accumulated gradient (1 / gradient_accumulation_steps), not of the pass
        -- with G x the rows per kernel launch.  The returned loss / kl are the means over the pass's groups.
        ``advantages`` may be a CALLABLE returning the list: it is invoked after both forward passes have been queued on the
        stream, so host-side reward functions (decode text -> regex / map scoring, TR:576-593) run while the GPU scores --
        the reference leaves the GPU idle there."""
        cfg, Gn = self.cfg, len(prompts)
        Kn, C = completions[0].shape
        # the rollouts of this step are done: the fragment-major decode copies of the weights (1x the LLM, 14 GB at 7B) are
        # rebuilt after the optimizer step anyway, so their memory goes to the scoring passes
        self.roll.invalidate()
        comp_all = torch.cat(completions, 0)
        mask, lengths = K.completion_mask(comp_all, cfg.eos_token_id)
        lens = lengths.tolist() if self.h.trim_completions else None      # (host read: the rollouts are finished by now)
        entries = [(p.ids, p.pix, p.grids) for p in prompts]
        pf = self._take_prefill(prompts, era_rule)
        with torch.no_grad():
            pr = self.h.precise_logps
            ref_lp = self.ref_engine.score_groups(entries, completions, era_rule=era_rule, precise=pr, lengths=lens)
            tape: dict = {}
            lp = self.engine.score_groups(entries, completions, tape=tape, era_rule=era_rule, precise=pr, prefill=pf, lengths=lens)
            del pf                                     # (their share of the kept tape is consumed; the last reference frees it)
            if callable(advantages):
                advantages = advantages()
            # the loss kernel averages over its rows: G groups of K rows -> mean over G*K rows = (1/G) * sum of group means
            loss, kl, dlogp = K.grpo_loss(lp, ref_lp, torch.cat(list(advantages)).to(self.dev), mask, self.h.beta)
            if grad_scale * Gn != 1.0:
                dlogp.mul_(grad_scale * Gn)
            hook = self.reducer.ready if (last_group and self.reducer is not None and self.h.overlap_comm) else None
            self.engine.backward_group(tape, dlogp, self.G, on_ready=hook)
        return dict(loss=loss, kl=kl, logps=lp, ref_logps=ref_lp, mask=mask, lengths=lengths, scored_tokens=int(sum(lens)) if lens is not None else mask.numel())

    def sft_forward_backward(self, ids: torch.Tensor, pix, grids, label_mask: torch.Tensor, *, grad_scale: float = 1.0,
                             second_per_grid_ts=None, last_group: bool = False) -> float:
        """Supervised objective of open_r1/sft.py on the same kernels: mean cross-entropy of ids[1:] over the positions whose
        label is kept (``label_mask`` bool/0-1 [S], True where labels != -100: the reference masks pad and visual tokens,
        sft.py:170-181), gradients accumulated into G scaled by grad_scale.  Returns the loss."""
        tape = {}
        logp = self.engine.score_sequence(ids, pix, grids, tape=tape, second_per_grid_ts=second_per_grid_ts)
        m = label_mask.reshape(-1)[1:].to(self.dev, torch.float32)
        n = float(m.sum().clamp_min(1.0))
        loss = float(-(logp * m).sum() / n)
        hook = self.reducer.ready if (last_group and self.reducer is not None and self.h.overlap_comm) else None
        self.engine.backward_group(tape, (-(m / n) * grad_scale).view(1, -1), self.G, on_ready=hook)
        return loss

    def reduce_gradients(self) -> None:
        """Data-parallel exchange: SUM all-reduce of the flat gradient over RCCL in large buckets (xGMI is per-link
        bound: few, big collectives).  Ranges already handed over during the last backward (``last_group=True``) are only
        waited for; the rest is sent now.  The mean is folded into the optimizer's grad_scale."""
        if self.reducer is not None:
            self.reducer.finish()
        if self.sharded is not None:
            t = self._comm_mark()
            self.sharded.reduce_scatter_(self.G.flat)
            self._comm_mark(t)

    def _comm_mark(self, start=None):
        """rs_ag runs on the compute stream: its collectives are exposed in full; bracket them with events when timing is on."""
        if not getattr(self, "_comm_on", False) or not self.G.flat.is_cuda:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        if start is not None:
            self._comm_events.append((start, e))
        return e

    def optimizer_step(self, world_size: int = 1) -> float:
        """Global-norm clip (max_grad_norm) + AdamW on fp32 master, bf16 policy refreshed in the same kernel."""
        h = self.h
        lr = lr_at(self.step_count, h)
        self.step_count += 1
        gscale = 1.0 / world_size
        K.zero_(self._sumsq)
        if self.sharded is not None:
            # rs_ag: this rank holds the summed gradient of ITS shard only -> global norm = all-reduce of the shards' sums of squares,
            # AdamW on the shard (master / moments of the other shards are never touched on this rank), all-gather of the bf16 weights
            import torch.distributed as dist
            sh = self.sharded
            K.sumsq_(sh.shard(self.G.flat), self._sumsq)
            tot = self._sumsq if sh.native else self._sumsq.cpu()
            dist.all_reduce(tot, group=self.pg)
            self._sumsq.copy_(tot)
            K.adamw_step_(sh.shard(self.master.flat), sh.shard(self.policy.flat), sh.shard(self.m), sh.shard(self.v), sh.shard(self.G.flat),
                          lr=lr, beta1=h.adam_beta1, beta2=h.adam_beta2, eps=h.adam_eps, weight_decay=h.weight_decay,
                          step=self.step_count, sumsq=self._sumsq, max_norm=h.max_grad_norm, grad_scale=gscale)
            t = self._comm_mark()
            sh.all_gather_(self.policy.flat)
            self._comm_mark(t)
        else:
            K.sumsq_(self.G.flat, self._sumsq)
            K.adamw_step_(self.master.flat, self.policy.flat, self.m, self.v, self.G.flat, lr=lr, beta1=h.adam_beta1,
                          beta2=h.adam_beta2, eps=h.adam_eps, weight_decay=h.weight_decay, step=self.step_count,
                          sumsq=self._sumsq, max_norm=h.max_grad_norm, grad_scale=gscale)
        K.zero_(self.G.flat)
        self.weights_changed()
        return lr

    def weights_changed(self) -> None:
        """EVERY writer of ``policy.flat`` calls this (the optimizer step above, a checkpoint load): a prefill tape kept by the rollout
        engine belongs to the weights that produced it, and the decode-layout weight copies are stale."""
        self.engine.weights_version += 1
        self.roll.invalidate()

    def gather_optimizer_state(self) -> None:
        """rs_ag keeps fp32 master weights and Adam moments current only in the owner's shard; before they are saved (or the
        algorithm is switched) every rank collects the other shards."""
        if self.sharded is not None:
            for t in (self.master.flat, self.m, self.v):
                self.sharded.all_gather_(t)
            self.sharded.drop_buffers(torch.float32)

    def grad_norm(self, world_size: int = 1) -> float:
        return float(self._sumsq.sqrt()) / world_size

    # -------------------------------------------------------------- data-parallel exchange accounting (bench.py ``comm``)
    def comm_timing(self, on: bool) -> None:
        """Start / stop collecting the exposed-exchange events (and reset the byte counter)."""
        self._comm_events = []
        if self.reducer is not None:
            self.reducer.timing, self.reducer.exposed, self.reducer.wire_bytes = on, [], 0
        self._comm_on = on

    def comm_stats(self, steps: int) -> Optional[dict]:
        """{algo, world, wire dtype, bytes handed to the collectives per step, bytes each GPU puts on its xGMI links per step
        (ring / direct all-reduce: 2 (n-1)/n x buffer; reduce-scatter + all-gather: (n-1)/n x (gradient + weights)), exposed_ms =
        time per step the compute stream waited for the exchange (what the last backward did not hide)}."""
        if self.pg is None:
            return None
        import torch.distributed as dist
        n = dist.get_world_size(self.pg)
        wire = 2 if self.h.grad_comm_bf16 else 4
        numel = self.G.flat.numel()
        torch.cuda.synchronize()
        if self.reducer is not None:
            ev = self.reducer.exposed
            per_step = self.reducer.wire_bytes / max(1, steps)
            on_wire = 2.0 * (n - 1) / n * numel * wire
            algo = "allreduce (bucketed, overlapped with the last backward)" if self.h.overlap_comm else "allreduce (after the last backward)"
        else:
            ev = getattr(self, "_comm_events", [])
            per_step = numel * wire + numel * 2
            on_wire = (n - 1) / n * (numel * wire + numel * 2)
            algo = "rs_ag (reduce-scatter + sharded AdamW + all-gather of bf16 weights)"
        ms = [a.elapsed_time(b) for a, b in ev]
        return {"algo": algo, "rccl_world": n, "wire_dtype": "bf16" if wire == 2 else "fp32", "bytes_to_collectives_per_step": round(per_step),
                "bytes_on_wire_per_gpu_per_step": round(on_wire), "exposed_ms": round(sum(ms) / max(1, steps), 3),
                "exposed_events": len(ms)}
