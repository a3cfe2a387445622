"""GPU, >= 2 devices (skipped on the one-GPU test box): the data-parallel exchange over RCCL ("nccl") with one process per GPU --
``allreduce_flat_`` and the overlapped ``GradReducer`` on device tensors (fp32 and bf16 wire), and one tiny two-rank GRPO step
whose replicas must end bit-identical (same reduced gradient, same AdamW)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    from spacer_amd.grpo import GradReducer, allreduce_flat_
    from spacer_amd.qwen2vl.config import TINY
    from spacer_amd.qwen2vl.weights import param_specs, total_numel
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    pg = dist.group.WORLD
    specs = param_specs(TINY)
    n = total_numel(specs)
    contrib = [torch.randn(n, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    ok = True
    for wire in (None, torch.bfloat16):
        want = sum(c.to(wire).float() if wire is not None else c for c in contrib)
        tol = dict(atol=3e-2, rtol=2e-2) if wire is not None else dict(atol=1e-5, rtol=1e-5)
        flat = contrib[rank].to(dev)
        allreduce_flat_(flat, pg, bucket_elems=100_000, wire_dtype=wire)
        ok = ok and torch.allclose(flat.cpu(), want, **tol)
        flat = contrib[rank].to(dev)
        red = GradReducer(flat, specs, pg, wire_dtype=wire, bucket_elems=100_000)
        red.ready("llm.norm_w"); red.ready("llm.lm_head")
        for i in reversed(range(TINY.layers)):
            red.ready(f"llm.{i}.")
        ok = ok and red.sent > 0
        red.finish()
        torch.cuda.synchronize()
        ok = ok and torch.allclose(flat.cpu(), want, **tol)
    # one data-parallel GRPO step on the tiny model: different prompts' completions per rank, identical replicas afterwards
    ok = ok and _dp_step(rank, world, pg, dev)
    ret[rank] = ok
    dist.destroy_process_group()


def _dp_step(rank, world, pg, dev):
    """Two optimizer steps of a data-parallel GRPO job under BOTH exchange algorithms (grpo.GRPOHyper.grad_algo): replicas
    bit-identical after every step, and the two algorithms agree with each other to the wire's rounding."""
    from golden_util import load_tiny
    from spacer_amd import kernels as K
    from spacer_amd.grpo import GRPOEngine, GRPOHyper, group_advantages
    from spacer_amd.qwen2vl.config import TINY
    from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict
    from spacer_amd.rollout import PromptInput, SamplingParams
    g = load_tiny()
    ok, finals = True, {}
    for algo in ("allreduce", "rs_ag"):
        params = FlatParams.empty(TINY, dev)
        load_state_dict(params, g["w"])
        ge = GRPOEngine(TINY, params, GRPOHyper(num_generations=3, learning_rate=1e-3, grad_algo=algo), process_group=pg)
        pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
        prompt = PromptInput(g["prompt"].to(dev), pix, [tuple(grid)])
        text_only = PromptInput(g["prompt"][-9:].to(dev), None, None)
        ge.comm_timing(True)
        for step in range(2):
            comp = ge.rollout([prompt], SamplingParams(max_new_tokens=8, seed=1 + rank + 10 * step))
            adv, _ = group_advantages(torch.tensor([2.0, 0.0, 1.0]), 3)
            # two micro-batches per step; the LAST one (whose backward feeds the overlapped reducer) is TEXT-ONLY on the odd ranks
            # and a video row on the even ones: every rank must still issue the same sequence of collectives
            ge.score_and_backward(prompt, comp, adv.to(dev), grad_scale=0.5)
            last = text_only if rank % 2 else prompt
            comp2 = ge.rollout([last], SamplingParams(max_new_tokens=8, seed=77 + rank + 10 * step))
            ge.score_and_backward(last, comp2, adv.to(dev), grad_scale=0.5, last_group=True)
            ge.reduce_gradients()
            ge.optimizer_step(world)
            torch.cuda.synchronize()
            mine = ge.policy.flat.float().cpu()
            both = [None] * world
            dist.all_gather_object(both, mine, group=pg)
            ok = ok and all(torch.equal(both[0], b) for b in both) and not torch.equal(mine, ge.ref.flat.float().cpu())
        st = ge.comm_stats(2)
        ok = ok and st["rccl_world"] == world and st["bytes_on_wire_per_gpu_per_step"] >= 0 and algo.split("_")[0] in st["algo"]
        if dist.get_backend(pg) == "nccl":
            ok = ok and st["exposed_events"] > 0 and st["exposed_ms"] >= 0.0      # the reducer's side stream / the sharded collectives were timed
        ge.gather_optimizer_state()
        masters = [None] * world
        dist.all_gather_object(masters, ge.master.flat.cpu(), group=pg)
        ok = ok and all(torch.equal(masters[0], m) for m in masters)        # after the gather the fp32 state is replicated too
        finals[algo] = ge.master.flat.cpu()
        del ge, params
    d = (finals["allreduce"] - finals["rs_ag"]).abs().max()
    return bool(ok and float(d) < 5e-3)          # same sums up to the bf16 wire's rounding order; lr 1e-3 x 2 steps bounds the drift


def _gloo_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)                 # both ranks share the box's one GPU; gradients are staged through the host
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ret[rank] = _dp_step(rank, world, dist.group.WORLD, dev)
    dist.destroy_process_group()


def test_data_parallel_step_both_exchange_algorithms_one_gpu():
    """The multi-rank control flow of a data-parallel step on a ONE-GPU box: two processes on cuda:0 over gloo."""
    ret = mp.Manager().dict()
    mp.spawn(_gloo_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
def test_rccl_gradient_exchange_two_gpus():
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _nccl_world1_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    from spacer_amd.grpo import GradReducer, ShardedExchange, allreduce_flat_
    from spacer_amd.qwen2vl.config import TINY
    from spacer_amd.qwen2vl.weights import param_specs, total_numel
    pg = dist.group.WORLD
    specs = param_specs(TINY)
    n = total_numel(specs)
    ok = True
    for wire in (None, torch.bfloat16):
        src = torch.randn(n, generator=torch.Generator().manual_seed(3))
        want = src.to(wire).float() if wire is not None else src
        flat = src.to(dev)
        allreduce_flat_(flat, pg, bucket_elems=100_000, wire_dtype=wire)
        ok = ok and torch.equal(flat.cpu(), want)
        flat = src.to(dev)
        red = GradReducer(flat, specs, pg, wire_dtype=wire, bucket_elems=100_000)      # RCCL collectives on the side stream
        red.ready("llm.norm_w"); red.ready("llm.lm_head")
        for i in reversed(range(TINY.layers)):
            red.ready(f"llm.{i}.")
        red.finish()
        torch.cuda.synchronize()
        ok = ok and torch.equal(flat.cpu(), want)
        flat = src.to(dev)
        sh = ShardedExchange(n, pg, wire_dtype=wire, bucket=70_000)                    # reduce_scatter_tensor / all_gather_into_tensor
        sh.reduce_scatter_(flat)
        sh.all_gather_(flat)
        torch.cuda.synchronize()
        ok = ok and torch.equal(flat.cpu(), want)
    ok = ok and _dp_step(0, 1, pg, dev)
    ret[0] = ok
    dist.destroy_process_group()


def test_rccl_code_path_on_one_gpu():
    """RCCL itself ("nccl" backend) with a world of ONE rank on the box's single GPU: the collectives are trivial, but every call the
    multi-GPU job makes -- process-group creation with device_id, async all_reduce on the reducer's side stream with bf16 wire
    buffers, reduce_scatter_tensor / all_gather_into_tensor of the sharded exchange, the scalar all-reduce of the clip norm, the
    data-parallel step under both algorithms -- runs against the real library on device buffers (the >= 2-GPU variant above skips
    on this box)."""
    ret = mp.Manager().dict()
    mp.spawn(_nccl_world1_worker, args=(1, _free_port(), ret), nprocs=1, join=True)
    assert ret[0]
