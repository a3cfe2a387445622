"""GPU: the driver's bench.py contract on the tiny workload -- one JSON line on stdout with the required keys (metric, value,
unit, n_gpus, steps, warmup, ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload) plus the
`roofline` and `cpu_baseline` objects; `--gpus N` beyond the visible GPUs refuses loudly instead of silently running one rank."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract_tiny(dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-pmc"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "samples/s" and d["value"] > 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["rccl_world"] == 1
    assert abs(d["value"] - d["config"]["global_batch"] * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("mfma", "hbm") and rf["unit"] in ("TFLOP/s", "GB/s") and rf["peak"] > 0 and "frac" in rf and "traffic" in rf
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "samples/s" and cb["sample"]
    v = d["variants"]
    assert set(v) == {"temporal", "free_running", "ragged", "through_trainer", "precise_scoring", "decode_cfg4_rows", "precise_step"}, set(v)
    rg = v["ragged"]        # EOS-trimmed scoring vs the [K, C] rectangle on the same seeded variable-length rollouts
    assert rg["trimmed"]["completion_tokens_scored_per_step"] < rg["rectangular"]["completion_tokens_scored_per_step"] == rg["trimmed"]["rectangle_tokens_per_step"]
    assert rg["trimmed"]["samples_per_s"] > 0 and rg["rectangular"]["samples_per_s"] > 0 and rg["speedup_trimmed_vs_rectangular"] > 0
    assert all("samples_per_s" in v[k] for k in ("temporal", "free_running", "through_trainer", "precise_step")), v
    assert v["through_trainer"]["gradient_accumulation_steps"] == 2 and v["through_trainer"]["vs_headline"] > 0
    assert v["through_trainer"]["steps"] == 4 and v["precise_step"]["vs_headline_step_time"] > 0
    assert d["config"]["precise_logps"] is False and d["config"]["groups_per_gpu"] == 2 and "launch_shape" in d["config"]
    rh = d["roofline_hbm"]
    assert rh["bound"] == "hbm" and rh["unit"] == "GB/s" and rh["achieved"] > 0 and rh["avg_launch_us"] > 0 and "skinny" in rh["kernel"]
    assert "value_kind" in cb
    assert v["precise_scoring"]["ratio"] > 0 and v["decode_cfg4_rows"]["ms_per_token_step"] > 0


def test_bench_refuses_more_gpus_than_visible(dev):
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--gpus", str(n)], capture_output=True,
                       text=True, timeout=300, env=env)
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_bench_two_ranks_control_flow_on_one_gpu(dev, algo):
    """`python bench.py --gpus 2` with no launcher environment becomes the launcher (torch.distributed.run, 2 ranks); with
    --backend gloo both ranks share cuda:0 and the overlapped gradient reducer stages through the host, so the whole multi-rank
    control flow -- rendezvous, per-rank seeds, reducer hooks during the last backward, barrier + max-over-ranks timing, rank 0
    printing ONE line with n_gpus = the world formed -- runs on the one-GPU test box (RCCL itself needs >= 2 GPUs:
    tests/test_multigpu_gpu.py)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--backend", "gloo", "--grad-algo", algo], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_world"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["grad_algo"] == algo
    assert d["config"]["global_batch"] == 2 * 2 * 4 and d["value"] > 0 and "cpu_baseline" not in d and "variants" not in d
    assert d["comm"]["rccl_world"] == 2 and d["comm"]["bytes_on_wire_per_gpu_per_step"] > 0 and algo.split("_")[0] in d["comm"]["algo"]


@pytest.mark.parametrize("groups,ragged", [(None, False), (1, False), (None, True)],
                         ids=["2-groups-per-rank", "cfg4-style-1-group-per-rank", "2-groups-per-rank-other-lengths-on-every-rank"])
@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_bench_eight_ranks_on_one_gpu(dev, algo, groups, ragged):
    """Multi-GPU readiness at the world size the driver's scaling run uses (run_SpaceR_SG_RLVR.sh:9-13 launches 8
    ranks, zero3.json:14-33 = overlapped bf16 gradient exchange): `bench.py --gpus 8 --backend gloo` = 8 processes on the box's ONE GPU
    (gradients staged through the host) -- shard bounds of the sharded exchange at n = 8, bucket schedule of the overlapped reducer, the
    packed metric gather, per-rank seeds, barrier + max-over-ranks timing.  Replicas must end bit-identical and the comm object must
    state 2 (n-1)/n x gradient bytes on the wire per GPU (all-reduce) / (n-1)/n x (gradient + weight bytes) (rs_ag), on the bf16 wire.
    With --groups 1 every rank holds ONE prompt group per step: the reference script's own launch shape (cfg4).  Round 6: with
    --synthetic-lengths every rank's rollouts END at other seeded lengths (EOS-trimmed scoring passes of other sizes on every rank: the
    collectives and the replicas must not care), and a weak-scaling line (groups > 1) carries the SECOND reading of the DP = 8 config --
    ``strong_cfg4`` (one group per rank, measured in the same job) -- plus the per-rank straggler spread; an N > 1 line never runs the
    variants, the CPU baseline or PMC passes (the driver runs N = 1, 2, 4, 8 back to back)."""
    from spacer_amd.qwen2vl.config import TINY
    from spacer_amd.qwen2vl.weights import param_specs, total_numel
    n = 8
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--gpus", str(n), "--steps", "2", "--warmup", "1",
           "--backend", "gloo", "--grad-algo", algo, "--check-replicas"] + (["--groups", str(groups)] if groups else []) \
        + (["--synthetic-lengths", "3,32"] if ragged else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    g = groups or 2
    assert d["n_gpus"] == n and d["config"]["rccl_world"] == n and d["config"]["parallelism"] == f"dp{n}" and d["config"]["grad_algo"] == algo
    assert d["config"]["groups_per_gpu"] == g and d["config"]["global_batch"] == n * g * 4 and d["value"] > 0
    assert d["replicas_identical"] is True
    numel = total_numel(param_specs(TINY))
    want = 2 * (n - 1) / n * numel * 2 if algo == "allreduce" else (n - 1) / n * (numel * 2 + numel * 2)
    assert d["comm"]["rccl_world"] == n and d["comm"]["wire_dtype"] == "bf16"
    assert d["comm"]["bytes_on_wire_per_gpu_per_step"] == round(want), (d["comm"], want)
    assert "variants" not in d and "cpu_baseline" not in d and "roofline_hbm" not in d and d["roofline"]["traffic"] is None
    assert set(d["n_gt_1_skips"]) >= {"variants", "cpu_baseline"}
    assert d["rank_compute_ms"]["max"] >= d["rank_compute_ms"]["min"] > 0
    if g > 1:        # the reference script's launch shape measured in the same job
        sc = d["strong_cfg4"]
        assert sc["groups_per_gpu"] == 1 and sc["global_batch"] == n * 4 and sc["value"] > 0 and sc["rccl_world"] == n
        assert sc["comm"]["bytes_on_wire_per_gpu_per_step"] == round(want) and sc["rank_compute_ms"]["min"] > 0
    else:
        assert "strong_cfg4" not in d
    if ragged:
        assert d["completion_tokens_scored_per_step"] < d["rectangle_tokens_per_step"] and "rollouts end" in d["config"]["workload"]
    else:
        assert d["completion_tokens_scored_per_step"] == d["rectangle_tokens_per_step"]


@pytest.mark.parametrize("algo", ["allreduce", "rs_ag"])
def test_bench_rccl_branch_with_a_world_of_one(dev, algo):
    """bench.py's distributed branch over RCCL ("nccl") on the one-GPU box: --force-dist forms a process group of one rank, so the
    process-group creation with device_id, the overlapped reducer on its side stream (bf16 wire), the barrier and the max-over-ranks
    timing all run against the real library -- the calls the driver's N = 2, 4, 8 runs make."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "2", "--warmup", "1", "--force-dist",
                        "--grad-algo", algo, "--no-variants", "--no-cpu-baseline", "--no-pmc"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "nccl" and d["config"]["grad_algo"] == algo and d["value"] > 0
    # the exchange's exposed time is measured with HIP events on the compute stream (world of one: nothing on the wire)
    assert d["comm"]["rccl_world"] == 1 and d["comm"]["exposed_events"] > 0 and d["comm"]["exposed_ms"] >= 0 and d["comm"]["bytes_on_wire_per_gpu_per_step"] == 0
