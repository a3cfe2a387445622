"""Dedicated rollout GPU: the topology of the reference's vLLM trainer, one process per GPU over RCCL / xGMI.

Reference: ``Qwen2VLGRPOVLLMTrainerModified`` (open_r1/trainer/vllm_grpo_trainer_modified.py).  There the MAIN training
process also drives a vLLM engine placed on the first GPU after the training ones (``vllm_device="auto"`` ->
``cuda:{num_processes}``, :325-327), and every step
  (1) loads the current policy ``state_dict`` into that engine when the global step moved (:526-545),
  (2) gathers every rank's prompt text + frames on the main process (``gather_object``, :548-549),
  (3) generates ``n = num_generations`` completions per prompt in ONE call with prefix caching (:565-576; the T-GRPO twin
      with n/2, :581-593),
  (4) broadcasts all completion ids to every rank, which slices out its own (:604-609).

MI355X shape of the same thing (this file): the rollout engine is its own RANK -- the last one of the job, with its own GPU
and only the bf16 policy + KV caches resident -- and the four steps become
  (1) a SHARDED weight push: the N training replicas hold identical weights, so replica r sends only slice r of the flat
      bf16 parameter buffer; the rollout rank receives N slices concurrently, one per point-to-point xGMI link (16.6 GB
      over 7 links instead of one), straight into its own flat buffer (no state_dict, no host copy);
  (2) a header gather (shapes only) + point-to-point sends of the prompt ids and the patchified bf16 pixel rows;
  (3) one ``RolloutEngine.generate`` over ALL ranks' prompts: decode streams the weights once per token for the whole job's
      rows (the shared-prompt decode attention is this engine's prefix cache);
  (4) a scatter: every rank receives only its own [prompts * K, C] ids.
The protocol is synchronous like the reference's (rollouts of step t are sampled from the weights of step t).

Only torch.distributed plumbing lives here; backend "nccl" (= RCCL) moves device tensors, "gloo" (CPU tests) is staged
through host memory by ``_wire``.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

from .rollout import PromptInput, SamplingParams

OP_SHUTDOWN, OP_GENERATE = 0, 1


@dataclass
class RolloutTopology:
    """ranks 0..N-1 train, rank N (the last) only generates -- the reference's "next GPU index" rule (:325-327)."""
    world_pg: object                 # group with all N + 1 ranks
    trainer_pg: object               # group of the N training ranks (None on the rollout rank's side of new_group is fine)
    rank: int
    world: int

    @property
    def server_rank(self) -> int:
        return self.world - 1

    @property
    def n_trainers(self) -> int:
        return self.world - 1

    @property
    def is_server(self) -> bool:
        return self.rank == self.server_rank


def make_topology(world_pg=None) -> RolloutTopology:
    """Collective over ALL ranks (``new_group`` must be entered by every rank, members or not)."""
    world_pg = world_pg if world_pg is not None else dist.group.WORLD
    world, rank = dist.get_world_size(world_pg), dist.get_rank(world_pg)
    if world < 2:
        raise ValueError("rollout-server mode needs at least 2 ranks (N trainers + 1 rollout rank); launch one more process than "
                         "training GPUs, as the reference asks for one more GPU than --num_processes")
    trainer_pg = dist.new_group(ranks=list(range(world - 1)))
    return RolloutTopology(world_pg, trainer_pg, rank, world)


def shard_bounds(numel: int, n: int, r: int, align: int = 512) -> tuple:
    """[lo, hi) of replica r's slice of a flat buffer split n ways on ``align``-element boundaries (last slice ragged)."""
    per = -(-numel // n)
    per = -(-per // align) * align
    return min(numel, r * per), min(numel, (r + 1) * per)


class _Wire:
    """Batched point-to-point transfers that work for both backends: RCCL takes device tensors and runs one batch as ONE
    grouped operation (all peers' links busy at once); gloo needs host tensors, and bf16 travels as int16 bits (not every
    gloo build has bf16).  Usage: queue ``send`` / ``recv_into`` calls, then ``flush()`` posts them together and waits."""

    def __init__(self, pg):
        self.pg = pg
        self.host = dist.get_backend(pg) == "gloo"
        self.ops: list = []
        self.after: list = []

    def send(self, t: torch.Tensor, dst: int) -> None:
        w = t.contiguous()
        if w.dtype == torch.bfloat16:
            w = w.view(torch.int16)
        if self.host:
            w = w.cpu()
        self.ops.append(dist.P2POp(dist.isend, w, dst, self.pg))

    def recv_into(self, t: torch.Tensor, src: int) -> None:
        assert t.is_contiguous()
        view = t.view(torch.int16) if t.dtype == torch.bfloat16 else t
        if self.host and view.device.type != "cpu":
            stage = torch.empty(view.shape, dtype=view.dtype)
            self.ops.append(dist.P2POp(dist.irecv, stage, src, self.pg))
            self.after.append(lambda: view.copy_(stage))
        else:
            self.ops.append(dist.P2POp(dist.irecv, view, src, self.pg))

    def flush(self) -> None:
        if self.ops:
            for w in dist.batch_isend_irecv(self.ops):
                w.wait()
        for f in self.after:
            f()
        self.ops, self.after = [], []


def _header_of(prompts: Sequence[PromptInput]) -> list:
    return [dict(n_ids=int(p.ids.numel()), pix=None if p.pix is None else tuple(p.pix.shape),
                 grids=None if p.grids is None else [tuple(int(v) for v in g) for g in p.grids],
                 sec=None if p.second_per_grid_ts is None else [float(v) for v in p.second_per_grid_ts]) for p in prompts]


# ------------------------------------------------------------------------------------------------ training-rank side
class RolloutClient:
    """What a training rank holds instead of a local RolloutEngine."""

    def __init__(self, topo: RolloutTopology, policy_flat: torch.Tensor):
        assert not topo.is_server
        self.topo, self.flat = topo, policy_flat
        self.wire = _Wire(topo.world_pg)
        self._pushed_version: Optional[int] = None

    def generate(self, prompts: List[PromptInput], num_generations: int, sp: SamplingParams, *, weights_version: int) -> torch.Tensor:
        """Same result contract as ``RolloutEngine.generate``: int64 [len(prompts) * num_generations, C] on the policy's
        device.  ``weights_version`` = the trainer's global step: the weight slice is pushed when it moved (the
        reference's ``_last_loaded_step`` test, :526-545).  EVERY training rank must call this in the same step."""
        t = self.topo
        push = self._pushed_version != weights_version
        head = dict(op=OP_GENERATE, push=push, G=int(num_generations), sp=dataclasses.asdict(sp), prompts=_header_of(prompts))
        dist.gather_object(head, None, dst=t.server_rank, group=t.world_pg)
        if push:
            lo, hi = shard_bounds(self.flat.numel(), t.n_trainers, t.rank)
            if hi > lo:
                self.wire.send(self.flat[lo:hi], t.server_rank)
            self._pushed_version = weights_version
        for p in prompts:
            self.wire.send(p.ids.to(torch.int64), t.server_rank)
            if p.pix is not None:
                self.wire.send(p.pix, t.server_rank)
        self.wire.flush()
        out = torch.empty(len(prompts) * num_generations, sp.max_new_tokens, dtype=torch.int64, device=self.flat.device)
        self.wire.recv_into(out, t.server_rank)
        self.wire.flush()
        return out

    def shutdown(self) -> None:
        """Collective over the training ranks: releases the rollout rank's ``serve`` loop."""
        dist.gather_object(dict(op=OP_SHUTDOWN), None, dst=self.topo.server_rank, group=self.topo.world_pg)


# ------------------------------------------------------------------------------------------------ rollout-rank side
class RolloutServer:
    """The rollout rank: a flat bf16 policy buffer that the trainers overwrite slice-wise + anything with
    ``generate(prompts, G, sp)`` and ``invalidate()`` on top of it (``spacer_amd.rollout.RolloutEngine``)."""

    def __init__(self, topo: RolloutTopology, policy_flat: torch.Tensor, engine, device=None):
        assert topo.is_server
        self.topo, self.flat, self.engine = topo, policy_flat, engine
        self.dev = device if device is not None else policy_flat.device
        self.wire = _Wire(topo.world_pg)
        self.steps_served = 0
        self.pushes = 0

    def serve_once(self) -> bool:
        t = self.topo
        heads: list = [None] * t.world
        dist.gather_object(None, heads, dst=t.server_rank, group=t.world_pg)
        heads = heads[:t.n_trainers]
        if heads[0]["op"] == OP_SHUTDOWN:
            return False
        if len({h["push"] for h in heads}) != 1 or len({h["G"] for h in heads}) != 1:
            raise RuntimeError(f"training ranks disagree on the request: {[(h['push'], h['G']) for h in heads]}")
        if heads[0]["push"]:                                      # (1) all slices in flight at once, one per xGMI link
            for r in range(t.n_trainers):
                lo, hi = shard_bounds(self.flat.numel(), t.n_trainers, r)
                if hi > lo:
                    self.wire.recv_into(self.flat[lo:hi], r)
        prompts: List[PromptInput] = []
        counts = []
        for r, h in enumerate(heads):                             # (2) prompt tensors, in the order the clients send them
            counts.append(len(h["prompts"]))
            for ph in h["prompts"]:
                ids = torch.empty(ph["n_ids"], dtype=torch.int64, device=self.dev)
                self.wire.recv_into(ids, r)
                pix = None
                if ph["pix"] is not None:
                    pix = torch.empty(ph["pix"], dtype=torch.bfloat16, device=self.dev)
                    self.wire.recv_into(pix, r)
                prompts.append(PromptInput(ids=ids, pix=pix, grids=ph["grids"], second_per_grid_ts=ph["sec"]))
        self.wire.flush()
        if heads[0]["push"]:
            self.engine.invalidate()                              # packed decode weights / cached vision features are stale
            self.pushes += 1
        G, sp = heads[0]["G"], SamplingParams(**heads[0]["sp"])
        ids = self.engine.generate(prompts, G, sp)                # (3) the whole job's rows in one decode loop
        a = 0
        for r, n in enumerate(counts):                            # (4) every rank gets its own rows only
            self.wire.send(ids[a * G:(a + n) * G], r)
            a += n
        self.wire.flush()
        self.steps_served += 1
        return True

    def serve(self) -> int:
        while self.serve_once():
            pass
        return self.steps_served
