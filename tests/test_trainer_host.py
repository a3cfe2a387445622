"""CPU: the host side of the drop-in surface -- C-ABI symbols, flag parsing, prompt construction, reward plumbing,
reward shaping, metric aggregation and the data-parallel exchange (world_size 2 over gloo)."""
import hashlib
import os
import re
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import grpo_ref as GR
from spacer_amd import _lib
from spacer_amd.grpo import GRPOHyper, allreduce_flat_, group_advantages, length_bonus, lr_at, temporal_bonus
from spacer_amd.open_r1 import SG_RLVR as ENTRY
from spacer_amd.open_r1.config import parse_args
from spacer_amd.open_r1.trainer import SG_RLVR_trainer as T

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ----------------------------------------------------------------------------------------------- C-ABI
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "spacer_hip.h")).read()
    declared = set(re.findall(r"\b(spacer_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) > 30
    lib = _lib.load()                                  # fails loudly if the .so is not built
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/spacer_hip.h but not exported"
    assert declared == set(_lib.SIGNATURES) | set(_lib.OTHER_SYMBOLS), declared ^ (set(_lib.SIGNATURES) | set(_lib.OTHER_SYMBOLS))
    assert lib.spacer_version() >= 100
    # argument validation runs before any launch, so it can be exercised without a GPU
    rc = lib.spacer_gemm_bf16_nt(None, 0, None, 0, None, 0, 1, 1, 64, None, None)
    assert rc == -1 and b"null operand" in lib.spacer_last_error()


def test_plan_predicates_answer_zero_on_a_rejected_plan():
    """the predicate entry points (``!= 0 means fused``; tile 128 / 256) must never hand a negative error code to a caller's
    truth test -- a plan of the wrong size (stale binding) answers 0 with the error string set.  Pure host functions: no GPU."""
    import ctypes as C
    lib = _lib.load()
    good, bad = _lib.Plan(), _lib.Plan()
    bad.struct_bytes = C.sizeof(_lib.Plan) - 4
    assert lib.spacer_gemm_tile(5498, 3584, 3584, 1, C.byref(good)) in (128, 256)
    assert lib.spacer_gemm_swiglu_fused(5498, 18944, 3584, C.byref(good)) == 1
    for call in (lambda: lib.spacer_gemm_tile(5498, 3584, 3584, 1, C.byref(bad)),
                 lambda: lib.spacer_gemm_swiglu_fused(5498, 18944, 3584, C.byref(bad)),
                 lambda: lib.spacer_gemm_pair_fused(5498, 3584, 3584, 1, C.byref(bad)),
                 lambda: lib.spacer_gemm_pair_epilogue_fused(_lib.SPACER_PAIR_SWIGLU, 5498, 37888, 3584, 128, 1, C.byref(bad))):
        assert call() == 0 and b"struct_bytes" in lib.spacer_last_error()


def test_kernels_refuse_cpu_tensors():
    from spacer_amd import kernels as K
    with pytest.raises(K.SpacerError):
        K.gemm_nt(torch.zeros(64, 64, dtype=torch.bfloat16), torch.zeros(64, 64, dtype=torch.bfloat16))


# ----------------------------------------------------------------------------------------------- flags / prompts
def test_launch_script_flags_parse():
    sh = open(os.path.join(ROOT, "scripts", "run_SpaceR_SG_RLVR.sh")).read()
    body = sh[sh.index("-m spacer_amd.open_r1.SG_RLVR") + len("-m spacer_amd.open_r1.SG_RLVR"):]
    argv = [t for t in re.sub(r"\\\n", " ", body).replace('"', "").split() if t]
    argv = [a.replace("${MODEL:-Qwen/Qwen2.5-VL-7B-Instruct}", "Qwen/Qwen2.5-VL-7B-Instruct").replace("${DATASET:-SpaceR-151k.jsonl}", "d.jsonl") for a in argv]
    script, train, model = parse_args(argv)
    assert script.temporal is True and script.len_control is True and script.max_pixels == 401408
    assert train.num_generations == 8 and train.beta == 0.04 and train.max_grad_norm == 5 and train.learning_rate == 1e-6
    assert train.max_completion_length == 1024 and train.lr_scheduler_type == "cosine" and train.bf16 is True
    assert train.deepspeed == "local_scripts/zero3.json" and model.attn_implementation == "flash_attention_2"


def test_prompt_templates_are_the_reference_strings():
    h = lambda s: hashlib.sha256(s.encode()).hexdigest()  # noqa: E731
    # sha256 of the reference's literals (open_r1/SG-RLVR.py:293-318), computed in the authoring container
    assert h(ENTRY.QUESTION_TEMPLATE) == "c2cea5bded10a50275c9487557d94f48a4aa80bc5ef708765678e252f072bfcd"
    assert h(ENTRY.COGMAP_TEMPLATE) == "fd1bb11ef399edabbf3e3c03c29a89446dea31d72d231fec84e3f7556ca836f9"
    assert h(repr(sorted(ENTRY.TYPE_TEMPLATE.items()))) == "a9a4eacd16a748d72b47514bdf573c9afb5e82a3b71575ff2339edd4514f46c4"
    ENTRY.R.MAP_DATA["v1"] = {"cognitive_map": {"chair": [[1, 1]], "tv": [[2, 2]]}, "object_list": ["chair", "tv"]}
    row = dict(problem="How many chairs?", problem_type="multiple choice", options=["A. 1", "B. 2"], data_source="SR_dataset",
               path="/x/v1.mp4", data_type="video")
    msg = ENTRY.make_conversation_image_and_video_map(row)["prompt"]
    assert msg[0]["content"][0] == {"type": "video"} and "['chair', 'tv']" in msg[0]["content"][1]["text"]
    assert msg[0]["content"][1]["text"].startswith("Question: How many chairs?Options:\nA. 1\nB. 2\n\n")
    txt = T.qwen2vl_chat_template(msg)
    assert txt.startswith("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n<|vision_start|><|video_pad|><|vision_end|>Question:")
    assert txt.endswith("<|im_end|>\n<|im_start|>assistant\n")


def test_reward_kwargs_and_none_stripping():
    inputs = [dict(prompt=[{"role": "user", "content": [{"type": "video", "text": None}, {"type": "text", "text": "q", "video": None}]}],
                   path="/v.mp4", problem_type="numerical", solution="<answer>3</answer>", problem_id=7, data_type="video")]
    kw = T.repeat_columns(inputs, 3)
    assert set(kw) == {"path", "problem_type", "solution", "problem_id", "data_type"} and kw["problem_id"] == [7, 7, 7]
    cleaned = T.remove_none_from_data([dict(m, content=[dict(p) for p in m["content"]]) for m in inputs[0]["prompt"]])
    assert cleaned[0]["content"] == [{"type": "video"}, {"type": "text", "text": "q"}]
    assert T.is_conversational(inputs[0]) and not T.is_conversational({"prompt": "plain"})


# ----------------------------------------------------------------------------------------------- reward shaping
def test_reward_shaping_matches_oracle_restatement():
    g = torch.Generator().manual_seed(0)
    for trial in range(50):
        G = 8
        rpf = torch.stack([torch.rand(G, generator=g) * 2 * (torch.rand(G, generator=g) > 0.4), (torch.rand(G, generator=g) > 0.5).float()], 1)
        sh = torch.stack([torch.rand(G // 2, generator=g) * 2, torch.ones(G // 2)], 1)
        lens = torch.randint(200, 700, (G,), generator=g)
        mask = (torch.arange(700)[None, :] < lens[:, None]).int()
        for temporal in (True, False):
            r1, t1 = temporal_bonus(rpf, sh, temporal, True)
            r2, t2 = GR.temporal_bonus(rpf, sh, temporal, True)
            assert torch.equal(r1, r2) and t1 == t2
            a1 = length_bonus(r1, rpf, lens, True)
            a2 = GR.length_bonus(r2, rpf, mask, True)
            assert torch.equal(a1, a2)
            adv1, s1 = group_advantages(a1, G)
            adv2, s2 = GR.group_advantages(a2, G)
            assert torch.allclose(adv1, adv2, atol=0, rtol=0) and torch.equal(s1, s2)


def test_lr_schedule():
    h = GRPOHyper(learning_rate=1e-6, total_steps=100, lr_scheduler_type="cosine")
    assert lr_at(0, h) == 1e-6 and abs(lr_at(50, h) - 0.5e-6) < 1e-12 and lr_at(100, h) < 1e-12
    h2 = GRPOHyper(learning_rate=1.0, total_steps=10, warmup_steps=2, lr_scheduler_type="linear")
    assert lr_at(0, h2) == 0.5 and lr_at(1, h2) == 1.0 and lr_at(2, h2) == 1.0 and abs(lr_at(6, h2) - 0.5) < 1e-12


def test_shard_indices_cover_dataset():
    parts = [T.shard_indices(103, r, 4, seed=1, epoch=0) for r in range(4)]
    assert all(len(p) == 26 for p in parts) and set(sum(parts, [])) == set(range(103))
    assert T.shard_indices(103, 0, 4, seed=1, epoch=0) != T.shard_indices(103, 0, 4, seed=1, epoch=1)


# ----------------------------------------------------------------------------------------------- world_size 2 (gloo)
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _dp_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = dist.group.WORLD
    # gradient exchange: bucketed flat all-reduce == plain sum
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(10_000, generator=g)
    mine = flat.clone()
    allreduce_flat_(flat, pg, bucket_elems=3000)
    others = [torch.randn(10_000, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)]
    ok_grad = torch.allclose(flat, sum(others)) and not torch.equal(flat, mine)
    # bf16 wire: every rank ends with the same values, equal to the sum of the bf16-rounded contributions
    wire = mine.clone()
    allreduce_flat_(wire, pg, bucket_elems=3000, wire_dtype=torch.bfloat16)
    want = sum(o.to(torch.bfloat16) for o in others).float()
    ok_grad = ok_grad and torch.allclose(wire, want, atol=2e-2, rtol=2e-2) and (wire - sum(others)).abs().max() < 5e-2
    # packed metrics: rank 0 group all wrong, rank 1 group all correct
    rpf = torch.tensor([[0.0, 1.0]] * 4) if rank == 0 else torch.tensor([[1.9, 1.0]] * 4)
    rewards = rpf.sum(1)
    packed = T.pack_metrics(torch.tensor([10, 20, 30, 40]) + rank, rpf, rewards, float(rank), torch.zeros(4), 0.25 * (rank + 1))
    buf = [torch.zeros_like(packed) for _ in range(world)]
    dist.all_gather(buf, packed, group=pg)
    m = T.reduce_metrics(torch.stack(buf), ["accuracy_reward", "format_reward"], temporal=True)
    ret[rank] = (ok_grad, m)
    dist.destroy_process_group()


def test_data_parallel_exchange_world2():
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        ok, m = ret[r]
        assert ok
        assert m["all_wrong"] == 0.5 and m["all_correct"] == 0.5          # fraction of ranks (TR:663-672)
        assert abs(m["completion_length"] - 25.5) < 1e-6 and abs(m["kl"] - 0.375) < 1e-6
        assert abs(m["rewards/accuracy_reward"] - 0.95) < 1e-6 and m["rewards/format_reward"] == 1.0
        assert m["temporal_rewards"] == 0.5 and abs(m["reward"] - (1.0 + 2.9) / 2) < 1e-6


def _reducer_worker(rank, world, port, ret):
    """GradReducer (grpo.py) on gloo/CPU tensors: ranges reported in BACKWARD order (end of the flat buffer first, as the
    engine's hooks do) are bucketed and sent while "backward" continues; finish() covers what was never reported."""
    from spacer_amd.grpo import GradReducer
    from spacer_amd.qwen2vl.config import TINY_TIED
    from spacer_amd.qwen2vl.weights import param_specs, total_numel
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    specs = param_specs(TINY_TIED)
    n = total_numel(specs)
    ok = True
    for wire in (None, torch.bfloat16):
        flat = torch.randn(n, generator=torch.Generator().manual_seed(7 + rank))
        contributions = [torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
        red = GradReducer(flat, specs, dist.group.WORLD, wire_dtype=wire, bucket_elems=200_000)
        sent_before_finish = 0
        red.ready("llm.norm_w")
        for i in reversed(range(TINY_TIED.layers)):
            red.ready(f"llm.{i}.")
        sent_before_finish = red.sent                              # buckets already on the wire "during backward"
        red.ready("llm.embed"); red.ready("merger.")
        for i in reversed(range(TINY_TIED.vit_depth)):
            red.ready(f"vit.{i}.")
        # vit.patch_w deliberately NOT reported: finish() must still cover it
        red.finish()
        want = sum(c.to(wire).float() if wire is not None else c for c in contributions)
        tol = dict(atol=3e-2, rtol=2e-2) if wire is not None else dict(atol=1e-6, rtol=1e-6)
        ok = ok and torch.allclose(flat, want, **tol) and sent_before_finish > 0
        # second step with NO ready() calls at all (overlap off): one finish() reduces everything
        flat2 = contributions[rank].clone()
        red2 = GradReducer(flat2, specs, dist.group.WORLD, wire_dtype=wire, bucket_elems=200_000)
        red2.finish()
        ok = ok and torch.allclose(flat2, want, **tol)
    ret[rank] = ok
    dist.destroy_process_group()


def test_overlapped_gradient_reducer_world2():
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_reducer_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def _sharded_worker(rank, world, port, ret):
    """grad_algo = "rs_ag" (grpo.ShardedExchange) on gloo/CPU tensors: after the reduce-scatter rank r holds the sum of shard r; an
    "optimizer" that touches only the own shard followed by the all-gather leaves every replica with the same full vector, equal to
    what the all-reduce algorithm computes (to fp32 summation order on an fp32 wire, to bf16 rounding on a bf16 wire)."""
    from spacer_amd.grpo import ShardedExchange
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 10_007                                                     # not a multiple of world or of the 64-element shard granule
    out = {}
    for wire in (None, torch.bfloat16):
        contrib = [torch.randn(n, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
        grad = contrib[rank].clone()
        sh = ShardedExchange(n, dist.group.WORLD, wire_dtype=wire, bucket=1000)
        sh.reduce_scatter_(grad)
        want = sum(c.to(wire).float() if wire is not None else c for c in contrib)
        tol = dict(atol=3e-2, rtol=2e-2) if wire is not None else dict(atol=1e-6, rtol=1e-6)     # fp32: summation order only
        ok = torch.allclose(grad[sh.lo:sh.hi], want[sh.lo:sh.hi], **tol)
        weights = torch.zeros(n)
        weights[sh.lo:sh.hi] = -0.1 * grad[sh.lo:sh.hi]            # the sharded "optimizer step"
        sh.all_gather_(weights)
        ref = contrib[rank].clone()
        allreduce_flat_(ref, dist.group.WORLD, bucket_elems=3000, wire_dtype=wire)
        ok = ok and torch.allclose(weights, -0.1 * ref, **(dict(atol=5e-3, rtol=2e-2) if wire is not None else dict(atol=1e-6, rtol=1e-6)))
        out[str(wire)] = (ok, weights)
    ret[rank] = out
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_reduce_scatter_all_gather_exchange_leaves_identical_replicas(world):
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_sharded_worker, args=(world, port, ret), nprocs=world, join=True)
    for wire in ("None", "torch.bfloat16"):
        assert all(ret[r][wire][0] for r in range(world))
        for r in range(1, world):
            assert torch.equal(ret[0][wire][1], ret[r][wire][1])    # replicas bit-identical
