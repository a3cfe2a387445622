"""Dedicated-rollout-rank topology (spacer_amd/rollout_server.py; reference vllm_grpo_trainer_modified.py:317-391,526-609) over
gloo on CPU: sharded weight push, point-to-point prompt transfer, one generate call for the whole job, scatter of the ids.
The generator here is a stand-in whose output is a pure function of (weights, prompt ids, pixel rows, grid, row), so every
received id proves which weights and which prompt reached the rollout rank; the HIP RolloutEngine behind the same server is
exercised on the GPU (tests/test_rollout_server_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from spacer_amd.rollout import PromptInput, SamplingParams
from spacer_amd.rollout_server import RolloutClient, RolloutServer, make_topology, shard_bounds

NUMEL = 70_001          # not a multiple of anything: ragged last slice


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _weights(version: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(7 + version)
    return torch.randn(NUMEL, generator=g).to(torch.bfloat16)


def _prompt(rank: int, step: int, j: int) -> PromptInput:
    g = torch.Generator().manual_seed(1000 * rank + 10 * step + j)
    n = 5 + rank + 2 * j
    ids = torch.randint(0, 500, (n,), generator=g)
    pix = torch.randn(4 * (j + 1), 8, generator=g).to(torch.bfloat16) if (rank + j) % 2 == 0 else None
    return PromptInput(ids=ids, pix=pix, grids=None if pix is None else [(1, 2, 2 * (j + 1))],
                       second_per_grid_ts=None if pix is None else [0.5])


def _expected(flat: torch.Tensor, p: PromptInput, G: int, C: int, seed: int) -> torch.Tensor:
    key = int(flat.float().sum().item() * 16) + int(p.ids.sum()) + seed
    if p.pix is not None:
        key += int(p.pix.float().abs().sum().item() * 4) + sum(sum(g) for g in p.grids) + int(p.second_per_grid_ts[0] * 10)
    rows = torch.arange(G).view(G, 1) * 100 + torch.arange(C).view(1, C)
    return (rows + key) % 100_000


class FakeEngine:
    def __init__(self, flat):
        self.flat, self.invalidations, self.batches = flat, 0, []

    def invalidate(self):
        self.invalidations += 1

    def generate(self, prompts, G, sp):
        self.batches.append(len(prompts))
        return torch.cat([_expected(self.flat, p, G, sp.max_new_tokens, sp.seed) for p in prompts], 0)


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    topo = make_topology()
    G, C = 4, 6
    if topo.is_server:
        flat = torch.zeros(NUMEL, dtype=torch.bfloat16)
        eng = FakeEngine(flat)
        srv = RolloutServer(topo, flat, eng)
        served = srv.serve()
        ret[rank] = dict(served=served, pushes=srv.pushes, invalidations=eng.invalidations, batches=eng.batches,
                         final_equal=bool(torch.equal(flat, _weights(1))))
    else:
        # the trainer subgroup works as a data-parallel group of its own (gradient exchange stays among the trainers)
        if topo.n_trainers > 1:
            x = torch.tensor([float(rank + 1)])
            dist.all_reduce(x, group=topo.trainer_pg)
            assert float(x) == sum(range(1, topo.n_trainers + 1))
        flat = _weights(0).clone()
        cl = RolloutClient(topo, flat)
        ok = True
        # step 0: two accumulation micro-steps on the same weights (ONE push), second one with a T-GRPO style pair of prompts
        for step, (version, n_prompts) in enumerate([(0, 1), (0, 2), (1, 1)]):
            if version == 1:
                flat.copy_(_weights(1))                     # the optimizer step happened
            prompts = [_prompt(rank, step, j) for j in range(n_prompts)]
            sp = SamplingParams(max_new_tokens=C, seed=3 + step)
            out = cl.generate(prompts, G, sp, weights_version=version)
            want = torch.cat([_expected(_weights(version), p, G, C, sp.seed) for p in prompts], 0)
            ok = ok and out.dtype == torch.int64 and out.shape == (n_prompts * G, C) and bool(torch.equal(out, want))
        cl.shutdown()
        ret[rank] = dict(ok=ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_rollout_rank_protocol(world):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    for r in range(world - 1):
        assert ret[r]["ok"], f"training rank {r} received wrong completions"
    s = ret[world - 1]
    assert s["served"] == 3 and s["pushes"] == 2 and s["invalidations"] == 2      # weights moved once after the first load
    assert s["batches"] == [world - 1, 2 * (world - 1), world - 1]                # one generate call for the whole job per request
    assert s["final_equal"]                                                       # slices reassembled bit-exactly


def test_shard_bounds_tile_the_buffer():
    for numel in (1, 511, 512, 70_001, 8_291_375_616):
        for n in (1, 2, 7):
            cuts = [shard_bounds(numel, n, r) for r in range(n)]
            assert cuts[0][0] == 0 and cuts[-1][1] == numel
            assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(lo <= hi for lo, hi in cuts)
            assert all(lo % 512 == 0 for lo, hi in cuts if hi > lo)


def test_flags_of_the_vllm_trainer_parse():
    from spacer_amd.open_r1.config import parse_args
    _, targs, _ = parse_args(["--use_vllm", "true", "--vllm_device", "auto", "--vllm_gpu_memory_utilization", "0.7"])
    assert targs.use_vllm is True and targs.vllm_device == "auto" and targs.vllm_gpu_memory_utilization == 0.7
    from spacer_amd.open_r1.trainer import Qwen2VLGRPOVLLMTrainerModified, SGRLVRTrainer
    assert issubclass(Qwen2VLGRPOVLLMTrainerModified, SGRLVRTrainer)
