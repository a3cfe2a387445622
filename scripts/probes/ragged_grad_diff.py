"""Probe: per-tensor gradient difference of the EOS-trimmed pass vs the rectangular pass vs the rectangular pass repeated (tiny model)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_ragged_gpu import _tiny, _completions
from spacer_amd import kernels as K
from spacer_amd.qwen2vl.config import TINY
from spacer_amd.qwen2vl.engine import Qwen2VLEngine

dev = torch.device("cuda:0")
g, params, prompts = _tiny(dev)
eng = Qwen2VLEngine(TINY, params)
C = 16
comps = [_completions(TINY, 5, C, [1, 7, C // 2, C, None], 11, dev), _completions(TINY, 5, C, [C // 2, None, 2, 1, C - 1], 12, dev)]
entries = [(p.ids, p.pix, p.grids) for p in prompts]
mask, lengths = K.completion_mask(torch.cat(comps, 0), TINY.eos_token_id)
lens = lengths.tolist()
dlogp = torch.randn(10, C, generator=torch.Generator().manual_seed(5)).to(dev) * 0.5 * mask.float()
grads = []
for ln in (lens, None, None):
    with K.plan(gemm_no_split=1, gemm_tile=256):
        t = {}
        eng.score_groups(entries, comps, tape=t, lengths=ln)
        G = eng.W.like(torch.float32)
        eng.backward_group(t, dlogp, G)
    grads.append(G)
for name, (a, b) in (("trim vs rect", (0, 1)), ("rect vs rect", (1, 2))):
    rows = []
    for spec in grads[0].specs:
        x, y = grads[a][spec.name].float(), grads[b][spec.name].float()
        den = float(y.norm())
        if den > 0:
            rows.append((float((x - y).norm()) / den, spec.name, den, float((x - y).abs().max())))
    rows.sort(reverse=True)
    print(name)
    for r in rows[:12]:
        print("   %.3e  %-24s |g| %.3e  max abs diff %.3e" % r)
