#!/usr/bin/env python
"""GPU: where does the engine's log-prob error come from?  Runs the tiny golden model through Qwen2VLEngine.score_group
with a tape and through oracle/qwen2vl_engine_emul.py (same bf16 rounding points, CPU) and prints, tensor by tensor and layer
by layer, max |engine - emulator| (should be fp32-summation-order small if the emulator models the engine) next to
max |engine - fp32 oracle|."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from golden_util import load_tiny  # noqa: E402
from oracle import qwen2vl_engine_emul as E  # noqa: E402
from oracle import qwen2vl_fp32 as O  # noqa: E402
from spacer_amd import kernels as K  # noqa: E402
from spacer_amd.qwen2vl.config import TINY  # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine  # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    g = load_tiny()
    cfg = g["cfg"]
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    eng = Qwen2VLEngine(TINY, params)
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    rows, _ = O.patchify_frames(g["frames"], cfg)
    rows = rows.to(torch.bfloat16).float()
    grids = [tuple(grid)]
    prompt, comps = g["prompt"], g["completions"]
    P, (Kn, C) = prompt.numel(), comps.shape

    tape = {}
    lp = eng.score_group(prompt.to(dev), comps.to(dev), pix, grids, tape=tape).cpu()
    want = O.completion_logps(wb, cfg, prompt, comps, rows, grids)
    emu = E.completion_logps(wb, cfg, prompt, comps, rows, grids)
    print(f"logp: |engine-oracle| {float((lp - want).abs().max()):.3e}  |emul-oracle| {float((emu - want).abs().max()):.3e}  "
          f"|engine-emul| {float((lp - emu).abs().max()):.3e}")

    R = E.Rounder()
    vit_c = []
    ve = E.vit_forward(wb, cfg, rows, grids, R, collect=vit_c)
    vit_o = O.vit_forward(wb, cfg, rows, grids)
    ve_eng = eng.vit_forward(pix, grids).float().cpu()
    print(f"vit out: |engine-emul| {float((ve_eng - ve).abs().max()):.3e}  |engine-oracle| {float((ve_eng - vit_o).abs().max()):.3e}  "
          f"|emul-oracle| {float((ve - vit_o).abs().max()):.3e}  (max |out| {float(vit_o.abs().max()):.2f})")
    D = cfg["vit_dim"]
    for i, (te, tc) in enumerate(zip(tape["vit"]["blocks"], vit_c)):
        for name in ("x_in", "h", "qkv", "o", "x_mid", "h2", "f1", "a"):
            a, b = te[name].float().cpu(), tc[name]
            if name == "qkv":
                a = a.reshape(b.shape)
            print(f"  vit[{i}].{name:6s} |engine-emul| {float((a - b).abs().max()):.3e}   max|.| {float(b.abs().max()):.2f}")
    # LLM: sequence 0 = prompt + completion 0 -> engine rows [0:P] + [P:P+C]
    ids = torch.cat([prompt, comps[0]])
    e0 = O.embed_with_video(wb, cfg, ids, ve)
    pos3, _ = O.mrope_position_ids(ids.tolist(), grids, cfg)
    llm_c = []
    xf = E.llm_hidden(wb, cfg, e0, pos3, R, collect=llm_c)
    sel = torch.cat([torch.arange(P), P + torch.arange(C)])
    for i, (te, tc) in enumerate(zip(tape["llm"], llm_c)):
        for name in ("x_in", "h", "qkv", "o", "x_mid", "h2", "gu", "a"):
            a, b = te[name].float().cpu()[sel], tc[name]
            if name == "gu":      # engine layout [gate | up]
                pass
            print(f"  llm[{i}].{name:6s} |engine-emul| {float((a - b).abs().max()):.3e}   max|.| {float(b.abs().max()):.2f}")
    a = tape["x_final"].float().cpu()[sel]
    print(f"  x_final |engine-emul| {float((a - xf).abs().max()):.3e}   max|.| {float(xf.abs().max()):.2f}")
    # logits of the selected rows
    hn = E._bf(O.rms_norm(xf[P - 1:-1], wb["model.norm.weight"], cfg["rms_eps"]))
    lg = hn @ O.lm_head_weight(wb, cfg).float().t()
    lge = tape["logits"].float().cpu()[:C]
    print(f"  logits(seq 0) |engine-emul| {float((lge - lg).abs().max()):.3e}   max|.| {float(lg.abs().max()):.2f}")
    hs = tape["hsel"].float().cpu()[:C]
    print(f"  hsel(seq 0) |engine-emul| {float((hs - hn).abs().max()):.3e}   max|.| {float(hn.abs().max()):.2f}")


if __name__ == "__main__":
    main()
