"""A/B in one process on one box: the 17..64-row decode step with the post-attention RMSNorm riding on the o projection's last-arriving
workgroups + the gate|up epilogue (RolloutEngine.ln2_fold, round 6) against the separate norm launch.  7B, cfg3 prompts, K = 8,
alternating per repetition; ms per token-step from the HIP events around the decode loop.

    python scripts/probes/decode_ln2_ab.py [C]"""
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
print(f"| rows | separate norm launch: ms per token-step (median of 4, min-max) | norm on the o launch | ratio |\n|---:|---:|---:|---:|")
for nP in (8, 4, 3):
    prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(nP)]
    res = {False: [], True: []}
    for rep in range(5):
        for fold in (False, True):
            roll.ln2_fold = fold
            st = {}
            roll.generate(prompts, 8, SamplingParams(max_new_tokens=C, seed=1, suppress_eos=True), stats=st)
            torch.cuda.synchronize()
            a, b, c = st["events"][0]
            if rep:
                res[fold].append(b.elapsed_time(c) / st["decode_steps"])
    cell = lambda v: f"{statistics.median(v):.3f} ({min(v):.3f}-{max(v):.3f})"      # noqa: E731
    print(f"| {nP * 8} | {cell(res[False])} | {cell(res[True])} | {statistics.median(res[True]) / statistics.median(res[False]):.4f} |", flush=True)
