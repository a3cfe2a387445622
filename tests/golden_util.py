"""Helpers to read tests/golden/tiny_model.npz (made by scripts/make_golden_model.py from HF transformers)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_tiny(name: str = "tiny_model.npz"):
    z = np.load(os.path.join(GOLDEN, name))
    cfg = json.loads(bytes(z["cfg"]).decode())
    cfg["mrope_section"] = tuple(cfg["mrope_section"])
    if "vit_fullatt" in cfg:
        cfg["vit_fullatt"] = tuple(cfg["vit_fullatt"])
    w = {k[3:]: torch.from_numpy(z[k].astype(np.float32)) for k in z.files if k.startswith("w::")}
    t = lambda k: torch.from_numpy(z[k])  # noqa: E731
    extra = {k: t(k) for k in ("hf_logps_sec2", "hf_pos_sec2", "hf_window_index", "hf_cu_window") if k in z.files}
    return dict(extra, cfg=cfg, w=w, frames=t("frames"), grid=tuple(int(v) for v in z["grid"]), prompt=t("prompt").long(),
                completions=t("completions").long(), hf_vit=t("hf_vit"), hf_logits_row0=t("hf_logits_row0"),
                hf_logps=t("hf_logps"), hf_pos=t("hf_pos").long(), hf_delta=int(z["hf_delta"]))


def load_tiny25():
    """tests/golden/tiny25_model.npz: the Qwen2.5-VL miniature (scripts/make_golden_model25.py)."""
    return load_tiny("tiny25_model.npz")
