#!/usr/bin/env python
"""BASELINE.json configs[0] executed in full on the host: Qwen2-VL-2B random-init (seed 1234, N(0, 0.02), tied lm_head),
2 prompts x 4 frames (280x364 -> 260 video tokens) + 360 text tokens, K = 2 rollouts of C tokens, the whole SG-RLVR step
per prompt group through oracle/cpu_path.py (ViT + prefill, KV-cache decode with top-k/top-p sampling, reference + policy
scoring, GRPO loss, autograd backward), fp32, torch CPU.  No GPU, no HIP library: this is the "CPU eager reference
(plumbing)" case.  Prints one JSON line.     usage: python scripts/run_cfg1_cpu.py [--C 32] [--threads N] [--layers L]"""
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
ap.add_argument("--C", type=int, default=32)
ap.add_argument("--threads", type=int, default=min(32, os.cpu_count() or 8))
ap.add_argument("--layers", type=int, default=28)
ap.add_argument("--vit-depth", type=int, default=32)
ap.add_argument("--prompts", type=int, default=2)
args = ap.parse_args()
torch.set_num_threads(args.threads)
cfg = O.make_config(hidden=1536, layers=args.layers, heads=12, kv_heads=2, intermediate=8960, vocab=151936, vit_dim=1280,
                    vit_depth=args.vit_depth, vit_heads=16, vit_mlp=5120, head_dim=128, tie_embeddings=True)
t0 = time.time()
w_ref = {k: v.float() for k, v in O.random_weights(cfg, seed=1234, dtype=torch.bfloat16).items()}
t_init = time.time() - t0
Kn, F, Hpx, Wpx, n_text = 2, 4, 280, 364, 360
res, t_all = [], time.time()
for p in range(args.prompts):
    g = torch.Generator().manual_seed(1000 + p)
    frames = torch.randint(0, 256, (F, 3, Hpx, Wpx), generator=g, dtype=torch.uint8)
    rows, grid = O.patchify_frames(frames, cfg)
    nv = grid[0] * grid[1] * grid[2] // 4
    text = torch.randint(1000, 150000, (n_text,), generator=torch.Generator().manual_seed(2000 + p))
    prompt = torch.cat([torch.tensor([151652]), torch.full((nv,), cfg["video_token_id"]), torch.tensor([151653]), text])
    w = {k: v.clone().requires_grad_(True) for k, v in w_ref.items()}
    out = CP.grpo_group_step(w, w_ref, cfg, prompt, rows, [tuple(grid)], num_generations=Kn, max_new_tokens=args.C, seed=p)
    res.append(out)
    print(f"prompt {p}: P={prompt.numel()} loss {out['loss']:.3e} seconds {json.dumps({k: round(v, 2) for k, v in out['seconds'].items()})}",
          file=sys.stderr, flush=True)
    del w
wall = time.time() - t_all
secs = {k: sum(r["seconds"][k] for r in res) for k in res[0]["seconds"]}
print(json.dumps({"config": f"cfg1: Qwen2-VL-2B random-init fp32 CPU, {args.prompts} prompts x {F} frames {Hpx}x{Wpx} x K={Kn}, C={args.C}, "
                            f"layers {args.layers}/vit {args.vit_depth}", "threads": args.threads, "init_seconds": round(t_init, 1),
                  "step_seconds": round(wall, 1), "samples_per_s": round(args.prompts * Kn / wall, 5),
                  "decode_tokens_per_s": round(args.prompts * Kn * (args.C - 1) / secs["decode"], 3),
                  "phase_seconds": {k: round(v, 2) for k, v in secs.items()},
                  "finite": all(bool(torch.isfinite(r["logps"]).all()) for r in res)}))
