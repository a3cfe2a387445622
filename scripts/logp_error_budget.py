#!/usr/bin/env python
"""Per-operator error budget of the per-token log-probs (SG_RLVR_trainer.py:353-366) on the CPU.

    python scripts/logp_error_budget.py tiny            # the golden 2-layer fixture
    python scripts/logp_error_budget.py 2b [--layers N] # Qwen2-VL-2B architecture (tied embeddings), seeded random init

For each rounding class of oracle/qwen2vl_engine_emul.py: the max / rms |logp - fp32 oracle| when ONLY that class
rounds to bf16, when every class BUT that one rounds, with everything rounding (= this engine's numerics) and with
selected classes carried as hi+lo bf16 pairs (two-pass MFMA operands).  Prints a markdown table.
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import qwen2vl_engine_emul as E  # noqa: E402
from oracle import qwen2vl_fp32 as O  # noqa: E402


def bf16_weights(w):
    return {k: v.to(torch.bfloat16).float() for k, v in w.items()}


def case_tiny(name="tiny_model.npz"):
    from golden_util import load_tiny
    g = load_tiny(name)
    w = bf16_weights(g["w"])
    w["visual.patch_embed.proj.weight"] = w["visual.patch_embed.proj.weight"].reshape(g["cfg"]["vit_dim"], -1)
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    return g["cfg"], w, g["prompt"], g["completions"], rows.to(torch.bfloat16).float(), [tuple(grid)]


def case_2b(layers, vit_depth, n_text, C, Kn, seed=1234, hidden=1536, heads=12, kv=2, inter=8960, vocab=151936, tie=True):
    cfg = O.make_config(hidden=hidden, layers=layers, heads=heads, kv_heads=kv, intermediate=inter, vocab=vocab, vit_dim=1280,
                        vit_depth=vit_depth, vit_heads=16, vit_mlp=5120, head_dim=128, tie_embeddings=tie)
    w = O.random_weights(cfg, seed=seed, dtype=torch.bfloat16)
    w = {k: v.float() for k, v in w.items()}
    g = torch.Generator().manual_seed(seed + 1)
    frames = torch.randint(0, 256, (4, 3, 112, 140), generator=g, dtype=torch.uint8)
    rows, grid = O.patchify_frames(frames, cfg)
    nv = grid[0] * grid[1] * grid[2] // 4
    text = torch.randint(1000, 150000, (n_text,), generator=g)
    prompt = torch.cat([torch.tensor([151652]), torch.full((nv,), cfg["video_token_id"]), torch.tensor([151653]), text])
    comps = torch.randint(1000, 150000, (Kn, C), generator=g)
    return cfg, w, prompt, comps, rows.to(torch.bfloat16).float(), [tuple(grid)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("case", choices=("tiny", "tiny25", "2b", "7b"))
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--vit-depth", type=int, default=32)
    ap.add_argument("--text", type=int, default=200)
    ap.add_argument("--C", type=int, default=24)
    ap.add_argument("--K", type=int, default=2)
    ap.add_argument("--quick", action="store_true", help="only the all / none / split rows")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    if args.case == "tiny":
        cfg, w, prompt, comps, rows, grids = case_tiny()
    elif args.case == "tiny25":
        cfg, w, prompt, comps, rows, grids = case_tiny("tiny25_model.npz")
    elif args.case == "7b":
        cfg, w, prompt, comps, rows, grids = case_2b(args.layers, args.vit_depth, args.text, args.C, args.K, hidden=3584, heads=28,
                                                     kv=4, inter=18944, vocab=152064, tie=False)
    else:
        cfg, w, prompt, comps, rows, grids = case_2b(args.layers, args.vit_depth, args.text, args.C, args.K)
    t0 = time.time()
    with torch.no_grad():
        want = O.completion_logps(w, cfg, prompt, comps, rows, grids)
        print(f"# {args.case}: layers {cfg['layers']}, vit depth {cfg['vit_depth']}, P = {prompt.numel()}, K x C = {tuple(comps.shape)}, "
              f"oracle {time.time() - t0:.1f} s; logp range [{float(want.min()):.2f}, {float(want.max()):.2f}]", flush=True)

        def err(points, split=()):
            got = E.completion_logps(w, cfg, prompt, comps, rows, grids, points=points, split=split)
            d = (got - want).abs()
            return float(d.max()), float(d.pow(2).mean().sqrt())

        print("| rounding to bf16 | max abs err | rms err |")
        print("|---|---|---|")
        rows_out = [("nothing (emulator == oracle)", (), ())]
        rows_out.append(("every class (this engine)", E.ALL_POINTS, ()))
        if not args.quick:
            for p in E.ALL_POINTS:
                rows_out.append((f"only `{p}`", (p,), ()))
            for p in E.ALL_POINTS:
                rows_out.append((f"all but `{p}`", tuple(q for q in E.ALL_POINTS if q != p), ()))
        rows_out.append(("all, `final` as hi+lo pair", E.ALL_POINTS, ("final",)))
        rows_out.append(("all, `final`+`norm` as pairs", E.ALL_POINTS, ("final", "norm")))
        rows_out.append(("all, `final`+`norm`+`act`+`o` as pairs (every GEMM A operand)", E.ALL_POINTS, ("final", "norm", "act", "o", "vit_out")))
        rows_out.append(("all as pairs", E.ALL_POINTS, E.ALL_POINTS))
        for label, pts, sp in rows_out:
            mx, rms = err(pts, sp)
            print(f"| {label} | {mx:.2e} | {rms:.2e} |", flush=True)
        from oracle import qwen2vl_bf16_emul as B   # the REFERENCE's numerics: HF bf16 eager, bf16 residual stream and logits
        d = (B.completion_logps(w, cfg, prompt, comps, rows, grids) - want).abs()
        print(f"| reference-style bf16 eager (every op output, residual stream and logits in bf16) | {float(d.max()):.2e} | {float(d.pow(2).mean().sqrt()):.2e} |")
    print(f"# total {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main()
