"""Generate tests/golden/patchify_hf.npz: HF transformers' own Qwen2-VL pre-processing (rescale 1/255, CLIP mean/std,
temporal patch replication, merge-block-major patchify -- the code behind the reference's processor call,
SG_RLVR_trainer.py:417-425) run in THIS container on seeded uint8 images, next to the inputs.  An image is the 2-frame
clip of itself (temporal_patch_size = 2), which is what the video path does with every frame pair.

    python scripts/make_golden_patchify.py

transformers 5.15 without torchvision exposes the PIL/numpy backend (Qwen2VLImageProcessorPil); the script asserts that
oracle/qwen2vl_fp32.py:patchify_frames reproduces it exactly -- that pins the oracle's K1 restatement.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qwen2vl_fp32 as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "patchify_hf.npz")


def main():
    from transformers import Qwen2VLImageProcessor
    proc = Qwen2VLImageProcessor(do_resize=False)
    cfg = dict(patch=14, tpatch=2, merge=2)
    blob = {}
    for i, (h, w) in enumerate(((56, 84), (84, 140), (28, 28))):
        g = torch.Generator().manual_seed(40 + i)
        img = torch.randint(0, 256, (h, w, 3), generator=g, dtype=torch.uint8)
        out = proc(images=[img.numpy()], return_tensors="pt")
        pv, grid = out["pixel_values"], tuple(int(v) for v in out["image_grid_thw"][0])
        frames = img.permute(2, 0, 1)[None].repeat(2, 1, 1, 1).contiguous()          # the 2-frame clip of the image
        rows, gr = O.patchify_frames(frames, cfg)
        assert gr == grid and torch.equal(rows, pv), (gr, grid, float((rows - pv).abs().max()))
        blob[f"frames{i}"] = frames.numpy(); blob[f"pixel_values{i}"] = pv.numpy(); blob[f"grid{i}"] = np.array(grid)
    np.savez_compressed(OUT, **blob)
    print("oracle patchify == HF processor on", len(blob) // 3, "images; wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
