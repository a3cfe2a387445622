"""Helpers to read tests/golden/tiny_model.npz (made by scripts/make_golden_model.py from HF transformers)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tiny():
    z = np.load(os.path.join(GOLDEN, "tiny_model.npz"))
    cfg = json.loads(bytes(z["cfg"]).decode())
    cfg["mrope_section"] = tuple(cfg["mrope_section"])
    w = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    return dict(cfg=cfg, w=w, frames=t("frames"), grid=tuple(int(v) for v in z["grid"]), prompt=t("prompt").long(),
                completions=t("completions").long(), hf_vit=t("hf_vit"), hf_logits_row0=t("hf_logits_row0"),
                hf_logps=t("hf_logps"), hf_pos=t("hf_pos").long(), hf_delta=int(z["hf_delta"]))
