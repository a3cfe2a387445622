"""A/B in one process on one box: the decode step with the q|k|v finishing step folded into the shared-prefix attention's split launch
(RolloutEngine.qkv_fused, round 6) against the separate finishing launch.  7B, cfg3 prompts, K = 8, alternating per repetition; ms per
token-step from the HIP events around the hipGraph decode loop.

    python scripts/probes/decode_qkv_fused_ab.py [C]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import statistics
import torch
from spacer_amd.qwen2vl.config import QWEN2_VL_7B as cfg
from spacer_amd.qwen2vl.engine import Qwen2VLEngine
from spacer_amd.qwen2vl.weights import FlatParams, random_init_
from spacer_amd.rollout import RolloutEngine, SamplingParams
from spacer_amd.synthetic import make_prompt

dev = torch.device("cuda:0")
params = FlatParams.empty(cfg, dev); random_init_(params, seed=1234)
roll = RolloutEngine(Qwen2VLEngine(cfg, params))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 256
print("| rows | separate finishing launch: ms per token-step (median of 4, min-max) | folded into the attention | ratio |\n|---:|---:|---:|---:|")
for nP in (8, 12, 1):
    prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(nP)]
    res = {False: [], True: []}
    for rep in range(5):
        for fused in (False, True):
            roll.qkv_fused = fused
            st = {}
            roll.generate(prompts, 8, SamplingParams(max_new_tokens=C, seed=1, suppress_eos=True), stats=st)
            torch.cuda.synchronize()
            a, b, c = st["events"][0]
            if rep:
                res[fused].append(b.elapsed_time(c) / st["decode_steps"])
    cell = lambda v: f"{statistics.median(v):.3f} ({min(v):.3f}-{max(v):.3f})"      # noqa: E731
    print(f"| {nP * 8} | {cell(res[False])} | {cell(res[True])} | {statistics.median(res[True]) / statistics.median(res[False]):.4f} |", flush=True)
