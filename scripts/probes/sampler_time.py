"""Sampler: one workgroup per row vs the wide form (several workgroups per row for the two passes over the logits), 152 064 logits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spacer_amd import kernels as K
dev = torch.device("cuda:0")
V = 152064
for B in (8, 12, 64, 96, 128):
    lg = [torch.randn(B, V, device=dev) * 2.5 for _ in range(6)]      # rotate: 6 x B x 608 KB
    step = torch.zeros(1, dtype=torch.int32, device=dev)
    out = torch.empty(B, dtype=torch.int64, device=dev)
    ws = K.sample_workspace(B, V, dev)
    res = {}
    for name, w in (("narrow", None), ("wide", ws), ("narrow", None), ("wide", ws)):
        for i in range(6):
            K.sample_top_p(lg[i], step, out_ids=out, workspace=w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(20):
            for i in range(6):
                K.sample_top_p(lg[i], step, out_ids=out, workspace=w)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(name, []).append(e0.elapsed_time(e1) / 120 * 1e3)
    print(f"B = {B:4d}: narrow {min(res['narrow']):6.1f} us, wide {min(res['wide']):6.1f} us per call", flush=True)
