"""Decode token-step time of the 7B rollout loop as a function of the batch's row count (prompts x K): 64 / 80 / 96 / 112 / 128 rows.
Question (round 6): the T-GRPO twin rollouts need only K/2 generations per twin prompt (TR:473: num_return_sequences = G // 2), i.e. 96
rows instead of the 128 the uniform-K batch decodes today -- would 96 rows be cheaper?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacer_amd.qwen2vl.config import QWEN2_VL_7B as cfg
from spacer_amd.qwen2vl.engine import Qwen2VLEngine
from spacer_amd.qwen2vl.weights import FlatParams, random_init_
from spacer_amd.rollout import RolloutEngine, SamplingParams
from spacer_amd.synthetic import make_prompt

dev = torch.device("cuda:0")
params = FlatParams.empty(cfg, dev); random_init_(params, seed=1234)
roll = RolloutEngine(Qwen2VLEngine(cfg, params))
C = int(sys.argv[1]) if len(sys.argv) > 1 else 96
for nP in (8, 10, 12, 14, 16, 12, 8):
    prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(nP)]
    best = 1e9
    for rep in range(2):
        st = {}
        roll.generate(prompts, 8, SamplingParams(max_new_tokens=C, seed=1, suppress_eos=True), stats=st)
        torch.cuda.synchronize()
        a, b, c = st["events"][0]
        best = min(best, b.elapsed_time(c) / st["decode_steps"])
    print(f"rows {nP * 8:4d}: {best:.3f} ms per token-step (prefill {a.elapsed_time(b):.0f} ms)", flush=True)
