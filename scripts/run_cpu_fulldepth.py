#!/usr/bin/env python
"""SURVEY 8(d) "CPU baseline": ONE prompt group of the headline workload (BASELINE.json configs[2]: 16 frames 280x364 -> 1040 video
tokens + 360 text tokens, K = 8 rollouts) through the WHOLE SG-RLVR step on the host at FULL DEPTH and the real vocabulary, with
the completion length reduced to C = 32 (oracle/cpu_path.py: ViT + prefill, KV-cache decode with top-k / top-p sampling,
reference + policy scoring, GRPO loss, autograd backward; fp32 torch, random-init weights N(0, 0.02) seed 1234).  Prints one JSON
line with the MEASURED samples/s and decode tokens/s and, as a separate labelled field, the extrapolation to C = 512 (decode and
the scoring passes scaled by their token counts).  Test / baseline infrastructure: no GPU, no HIP library.

    python scripts/run_cpu_fulldepth.py --model 7b [--C 32] [--threads 32]      # ~100 GB of host RAM, minutes
    python scripts/run_cpu_fulldepth.py --model 2b"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu_path as CP  # noqa: E402
from oracle import qwen2vl_fp32 as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", choices=("2b", "7b"), default="7b")
ap.add_argument("--C", type=int, default=32)
ap.add_argument("--K", type=int, default=8)
ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
args = ap.parse_args()
torch.set_num_threads(args.threads)
if args.model == "7b":
    cfg = O.make_config(hidden=3584, layers=28, heads=28, kv_heads=4, intermediate=18944, vocab=152064, vit_dim=1280, vit_depth=32,
                        vit_heads=16, vit_mlp=5120, head_dim=128, tie_embeddings=False)
else:
    cfg = O.make_config(hidden=1536, layers=28, heads=12, kv_heads=2, intermediate=8960, vocab=151936, vit_dim=1280, vit_depth=32,
                        vit_heads=16, vit_mlp=5120, head_dim=128, tie_embeddings=True)
t0 = time.time()
w_ref = {k: v.float() for k, v in O.random_weights(cfg, seed=1234, dtype=torch.bfloat16).items()}
t_init = time.time() - t0
n_params = sum(v.numel() for v in w_ref.values())
Kn, C, F, Hpx, Wpx, n_text = args.K, args.C, 16, 280, 364, 360
g = torch.Generator().manual_seed(1000)
frames = torch.randint(0, 256, (F, 3, Hpx, Wpx), generator=g, dtype=torch.uint8)
rows, grid = O.patchify_frames(frames, cfg)
nv = grid[0] * grid[1] * grid[2] // 4
text = torch.randint(1000, 150000, (n_text,), generator=torch.Generator().manual_seed(2000))
prompt = torch.cat([torch.tensor([151652]), torch.full((nv,), cfg["video_token_id"]), torch.tensor([151653]), text])
P = prompt.numel()
w = {k: v.clone().requires_grad_(True) for k, v in w_ref.items()}
out = CP.grpo_group_step(w, w_ref, cfg, prompt, rows, [tuple(grid)], num_generations=Kn, max_new_tokens=C, seed=0)
sec = out["seconds"]
total = out["total_seconds"]
# extrapolation to the workload's C = 512: decode ~ per-token-step time x (512 - 1); the three scoring passes ~ token count
C_full = 512
tok_s, tok_f = P + Kn * C, P + Kn * C_full
full = {"vit+prefill": sec["vit+prefill"], "decode": sec["decode"] / max(1, C - 1) * (C_full - 1)}
for k in ("ref scoring", "policy scoring", "loss+backward"):
    full[k] = sec[k] * tok_f / tok_s
t_full = sum(full.values())
print(json.dumps({
    "config": f"one cfg3 prompt group on the host: Qwen2-VL-{args.model.upper()} full depth (28 + 32 layers, vocab {cfg['vocab']}), fp32, "
              f"{F} frames {Hpx}x{Wpx}, P={P}, K={Kn}, C={C}",
    "threads": args.threads, "params": n_params, "init_seconds": round(t_init, 1),
    "measured": {"group_seconds": round(total, 1), "samples_per_s": round(Kn / total, 5),
                 "decode_tokens_per_s": round(Kn * (C - 1) / sec["decode"], 3),
                 "phase_seconds": {k: round(v, 2) for k, v in sec.items()}},
    "extrapolated_to_C512": {"group_seconds": round(t_full, 1), "samples_per_s": round(Kn / t_full, 5),
                             "phase_seconds": {k: round(v, 1) for k, v in full.items()},
                             "rule": "decode x (511 / (C - 1)); scoring and backward x (P + K 512) / (P + K C); ViT + prefill unchanged"},
    "finite": bool(torch.isfinite(out["logps"]).all()), "loss": out["loss"]}))
