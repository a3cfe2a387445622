"""A/B of two builds of libspacer_hip.so inside ONE process on the 7B decode loop (64 rows): ms per token-step, alternating.
usage: decode_ab_libs.py <other .so> [C]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacer_amd import _lib
from spacer_amd.qwen2vl.config import QWEN2_VL_7B as cfg
from spacer_amd.qwen2vl.engine import Qwen2VLEngine
from spacer_amd.qwen2vl.weights import FlatParams, random_init_
from spacer_amd.rollout import RolloutEngine, SamplingParams
from spacer_amd.synthetic import make_prompt

def open_lib(path):
    saved, _lib._lib, _lib.LIB_PATH = _lib._lib, None, path
    lib = _lib.load()
    _lib._lib = saved
    return lib
here = _lib.LIB_PATH
libs = {"this tree": _lib.load(), "other": open_lib(sys.argv[1])}
_lib.LIB_PATH = here
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda:0")
params = FlatParams.empty(cfg, dev); random_init_(params, seed=1234)
roll = RolloutEngine(Qwen2VLEngine(cfg, params))
prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(8)]
res = {k: [] for k in libs}
for rep in range(4):
    for name, lib in libs.items():
        _lib._lib = lib
        st = {}
        roll.generate(prompts, 8, SamplingParams(max_new_tokens=Cn, seed=1, suppress_eos=True), stats=st)
        torch.cuda.synchronize()
        a, b, c = st["events"][0]
        res[name].append(b.elapsed_time(c) / st["decode_steps"])
for k, v in res.items():
    print(f"{k:10s}: ms per token-step {', '.join(f'{x:.3f}' for x in v)}   best {min(v):.3f}")
