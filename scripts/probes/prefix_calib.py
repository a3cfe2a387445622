import sys, torch
sys.path.insert(0, "/root/repo")
from spacer_amd.qwen2vl.config import QWEN2_VL_7B as cfg
from spacer_amd.qwen2vl.engine import Qwen2VLEngine
from spacer_amd.qwen2vl.weights import FlatParams, random_init_
from spacer_amd.synthetic import make_prompt
dev = torch.device("cuda:0")
params = FlatParams.empty(cfg, dev); random_init_(params, seed=7)
eng = Qwen2VLEngine(cfg, params)
prompt, _ = make_prompt(cfg, 0, 16, 280, 364, 360, dev)
comps = torch.randint(1000, 150000, (8, 512), generator=torch.Generator().manual_seed(3)).to(dev)
lp = eng.score_group(prompt.ids, comps, prompt.pix, prompt.grids)
lp2 = eng.score_group(prompt.ids, comps, prompt.pix, prompt.grids)
print("same call twice: max", float((lp - lp2).abs().max()))
for k in (0, 5):
    alone = eng.score_group(prompt.ids, comps[k:k + 1], prompt.pix, prompt.grids)
    d = lp[k] - alone[0]
    print(k, "max", float(d.abs().max()), "rms", float(d.pow(2).mean().sqrt()), "n>0.03", int((d.abs() > 0.03).sum()), "first 8 tok max", float(d[:8].abs().max()),
          "by quarter", [round(float(d[i*128:(i+1)*128].pow(2).mean().sqrt()), 4) for i in range(4)])
two = eng.score_group(prompt.ids, comps[:2], prompt.pix, prompt.grids)
d = lp[:2] - two
print("8-group vs 2-group: max", float(d.abs().max()), "rms", float(d.pow(2).mean().sqrt()))
