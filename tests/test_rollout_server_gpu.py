"""GPU: the dedicated-rollout-rank topology end to end on the tiny model -- a training rank running
``Qwen2VLGRPOVLLMTrainerModified`` and a rollout rank running the HIP RolloutEngine behind ``RolloutServer`` (two processes;
on a box with >= 2 GPUs each rank takes its own GPU and the job talks over RCCL ("nccl": grouped point-to-point weight push,
device tensors on the wire); on a one-GPU box both ranks share cuda:0 and talk over gloo, the same protocol staged through host
memory).  Bar: the remote rollouts are the ones a local RolloutEngine samples from the same weights and seed,
so step 1 of the remote trainer reproduces step 1 of ``SGRLVRTrainer`` exactly (rewards, lengths, loss)."""
import json
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


def content_reward(prompts, completions, video_path=None, **kw):
    return [float(sum(ord(ch) for ch in c[0]["content"]) % 5) / 2 for c in completions]


def _worker(rank, world, port, tmp, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPACER_SKINNY_BLOCKS="1")   # reproducible decode sums
    from fake_processor import FakeProcessor
    from golden_util import load_tiny
    from spacer_amd.open_r1.config import GRPOConfig, GRPOScriptArguments
    from spacer_amd.open_r1.rewards import format_reward
    from spacer_amd.open_r1.trainer import Qwen2VLGRPOVLLMTrainerModified, SGRLVRTrainer
    from spacer_amd.open_r1.trainer.vllm_grpo_trainer_modified import run_rollout_rank
    from spacer_amd.qwen2vl.config import TINY
    from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict
    from spacer_amd.rollout_server import make_topology
    multi = torch.cuda.device_count() >= world
    dev = torch.device("cuda", rank if multi else 0)
    torch.cuda.set_device(dev)
    if multi:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    ret["backend"] = dist.get_backend()
    topo = make_topology()
    if topo.is_server:
        ret["served"] = run_rollout_rank(TINY, topo, dev)
        dist.destroy_process_group()
        return
    g = load_tiny()
    rows = []
    for i in range(2):
        frames = torch.randint(0, 256, (6, 3, 56, 84), generator=torch.Generator().manual_seed(i), dtype=torch.uint8)
        rows.append(dict(prompt=[{"role": "user", "content": [{"type": "video"}, {"type": "text", "text": f"what is in clip {i} ?"}]}],
                         path=frames, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>",
                         problem_id=i, options=["A. x", "B. y"], data_source="other"))
    logs = {}
    for kind in ("local", "remote"):
        params = FlatParams.empty(TINY, dev)
        load_state_dict(params, g["w"])
        out_dir = os.path.join(tmp, kind)
        args = GRPOConfig(output_dir=out_dir, max_completion_length=8, num_generations=4, learning_rate=1e-4, max_steps=2,
                          logging_steps=1, save_steps=0, seed=3, use_vllm=(kind == "remote"))
        common = dict(model=params, reward_funcs=[content_reward, format_reward], args=args,
                      script_args=GRPOScriptArguments(temporal=True, len_control=True), train_dataset=rows,
                      processing_class=FakeProcessor(TINY), device=dev)
        tr = SGRLVRTrainer(**common) if kind == "local" else Qwen2VLGRPOVLLMTrainerModified(topology=topo, **common)
        tr.train()
        logs[kind] = [json.loads(line) for line in open(os.path.join(out_dir, "trainer_log.jsonl"))]
        logs[kind + "_moved"] = not torch.equal(tr.engine.policy.flat, tr.engine.ref.flat)
    ret["logs"] = logs
    dist.destroy_process_group()


def test_remote_rollouts_reproduce_the_local_trainer(tmp_path):
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), ret), nprocs=2, join=True)
    logs = ret["logs"]
    assert ret["served"] == 2                      # one request per optimizer step (prompt + T-GRPO twin in the same request)
    assert logs["local_moved"] and logs["remote_moved"]
    a, b = logs["local"], logs["remote"]
    assert len(a) == len(b) == 2
    # step 1: same weights, same seed -> the rollout rank samples exactly the local engine's completions
    for key in ("completion_length", "rewards/content_reward", "rewards/format_reward", "reward", "reward_std", "temporal_rewards"):
        assert a[0][key] == b[0][key], (key, a[0][key], b[0][key])
    assert abs(a[0]["loss"] - b[0]["loss"]) < 1e-6 and abs(a[0]["kl"] - b[0]["kl"]) < 1e-6
    # step 2 runs on the pushed (updated) weights; backward atomics make the update reproducible to rounding only
    assert abs(a[1]["loss"] - b[1]["loss"]) < 5e-2
