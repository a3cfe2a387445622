"""The decode loop of the cfg3 workload (Qwen2-VL-7B, 8 prompts x K = 8 = 64 rows, 16 frames) for a few token steps, launched
eagerly (no hipGraph) so that rocprofv3 --pmc attributes counters to every kernel of a decode step:

    cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE -d <dir> -o r -- python scripts/decode_pmc.py [steps] [groups]
    python scripts/pmc_kernel_bw.py <dir>/.../r_results.db profiles/r03_decode_pmc.md "<title>" gemm_skinny attn_decode decode_ norm_fwd sample embed
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spacer_amd.qwen2vl.config import PRESETS                   # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine             # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, random_init_  # noqa: E402
from spacer_amd.rollout import RolloutEngine, SamplingParams    # noqa: E402
from spacer_amd.synthetic import make_prompt                    # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
groups = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
cfg = PRESETS["Qwen2-VL-7B"]
params = FlatParams.empty(cfg, dev)
random_init_(params, seed=1234)
roll = RolloutEngine(Qwen2VLEngine(cfg, params))
prompts = [make_prompt(cfg, g, 16, 280, 364, 360, dev)[0] for g in range(groups)]
sp = SamplingParams(max_new_tokens=steps + 1, seed=1, suppress_eos=True)
out = roll.generate(prompts, 8, sp, use_graph=False)
torch.cuda.synchronize()
print("decoded", tuple(out.shape))
