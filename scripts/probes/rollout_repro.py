"""Determinism stress for the rollout path on the tiny golden model: eager vs eager vs hipGraph, repeated with idle gaps
(cold clocks).  Prints mismatch counts; env toggles pick kernel variants.   usage: rollout_repro.py [iters]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_tiny                      # noqa: E402
from spacer_amd import kernels as K                    # noqa: E402
from spacer_amd.qwen2vl.config import TINY             # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine    # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict  # noqa: E402
from spacer_amd.rollout import PromptInput, RolloutEngine, SamplingParams  # noqa: E402

dev = torch.device("cuda:0")
g = load_tiny()
params = FlatParams.empty(TINY, dev)
load_state_dict(params, g["w"])
pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
prompts = [PromptInput(g["prompt"].to(dev), pix, [tuple(grid)]), PromptInput(g["prompt"][-9:].to(dev), None, None)]
eng = Qwen2VLEngine(TINY, params)
roll = RolloutEngine(eng)
sp = SamplingParams(max_new_tokens=12, top_k=50, top_p=0.95, seed=int(os.environ.get("SEED", "11")), suppress_eos=True)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
bad_ee = bad_eg = 0
ref = None
for it in range(iters):
    a = roll.generate(prompts, 3, sp, use_graph=False)
    b = roll.generate(prompts, 3, sp, use_graph=True)
    c = roll.generate(prompts, 3, sp, use_graph=False)
    if ref is None:
        ref = a.clone()
    ee = not torch.equal(a, c) or not torch.equal(a, ref)
    eg = not torch.equal(a, b)
    bad_ee += ee; bad_eg += eg
    if ee or eg:
        d = (a != b).nonzero().tolist()[:4], (a != c).nonzero().tolist()[:4], (a != ref).nonzero().tolist()[:4]
        print(f"  iter {it}: eager/graph diff at {d[0]}, eager/eager {d[1]}, vs first {d[2]}")
    time.sleep(0.7)
print(f"iters {iters}: eager-vs-eager mismatches {bad_ee}, eager-vs-graph mismatches {bad_eg}")
