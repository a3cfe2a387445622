"""Generate tests/golden/tiny25_model.npz by running HF transformers' Qwen2_5_VLForConditionalGeneration (what the
reference's shipped script trains: run_SpaceR_SG_RLVR.sh:16, SG_RLVR_trainer.py:184-190) in THIS container, random-init,
fp32, eager attention, on a tiny config that keeps the real head sizes (LLM 128, ViT 80), a ragged SwiGLU width (420),
56-pixel windows (2x2 merge units, ragged at the border) and alternating window / full-attention blocks.

    python scripts/make_golden_model25.py

Asserts that oracle/qwen2vl_fp32.py (vit_kind = "qwen2_5") reproduces HF to fp32 round-off -- this pins the oracle's
Qwen2.5-VL vision tower, window index and temporally scaled M-RoPE -- and stores config, weights, inputs and HF outputs.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qwen2vl_fp32 as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiny25_model.npz")

CFG = O.make_config(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=1024,
                    vit_dim=320, vit_depth=4, vit_heads=4, vit_mlp=420, head_dim=128,
                    video_token_id=1001, image_token_id=1000, tie_embeddings=False,
                    vit_kind="qwen2_5", vit_window=56, vit_fullatt=(1, 3), tokens_per_second=2)
VISION_START, VISION_END = 1002, 1003


def hf_model():
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration
    cfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6,
                         rope_parameters=dict(rope_theta=1e6, rope_type="default", mrope_section=[16, 24, 24]),
                         max_position_embeddings=4096, tie_word_embeddings=False),
        vision_config=dict(depth=4, hidden_size=320, out_hidden_size=256, num_heads=4, intermediate_size=420, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3, window_size=56,
                           fullatt_block_indexes=[1, 3], tokens_per_second=2, hidden_act="silu"),
        image_token_id=1000, video_token_id=1001, vision_start_token_id=1002, vision_end_token_id=1003,
        tie_word_embeddings=False,
    )
    cfg._attn_implementation = "eager"
    torch.manual_seed(9)
    m = Qwen2_5_VLForConditionalGeneration(cfg).float().eval()
    g = torch.Generator().manual_seed(13)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            p.copy_(p.half().float())
    return m


def to_ckpt_names(sd):
    out = {}
    for k, v in sd.items():
        if k.startswith("model.visual."):
            k2 = k[len("model."):]
        elif k.startswith("model.language_model."):
            k2 = "model." + k[len("model.language_model."):]
        else:
            k2 = k
        if k2 == "visual.patch_embed.proj.weight":
            v = v.reshape(v.shape[0], -1)
        out[k2] = v.detach().clone()
    return out


def main():
    m = hf_model()
    w = to_ckpt_names(m.state_dict())
    g = torch.Generator().manual_seed(5)
    frames = torch.randint(0, 256, (4, 3, 84, 140), generator=g, dtype=torch.uint8)   # grid (2, 6, 10): 3x5 merge units, 2x2 windows ragged
    rows, grid = O.patchify_frames(frames, CFG)
    nv = grid[0] * grid[1] * grid[2] // 4
    text = torch.randint(5, 990, (9,), generator=g)
    prompt = torch.cat([torch.tensor([VISION_START]), torch.full((nv,), 1001), torch.tensor([VISION_END]), text])
    comps = torch.randint(5, 990, (3, 6), generator=g)
    P = prompt.numel()

    ids = torch.stack([torch.cat([prompt, c]) for c in comps])
    K = ids.shape[0]
    mm = ((ids == 1001) * 2 + (ids == 1000) * 1).int()
    gthw = torch.tensor([grid] * K)
    with torch.no_grad():
        out = m(input_ids=ids, pixel_values_videos=rows.repeat(K, 1), video_grid_thw=gthw, mm_token_type_ids=mm)
        vit = m.model.visual(rows, grid_thw=torch.tensor([grid])).pooler_output
        sec = torch.tensor([2.0] * K)
        out2 = m(input_ids=ids, pixel_values_videos=rows.repeat(K, 1), video_grid_thw=gthw, mm_token_type_ids=mm,
                 second_per_grid_ts=sec)
    logits = out.logits.float()
    lp = torch.log_softmax(logits[:, :-1], -1).gather(2, ids[:, 1:, None]).squeeze(2)[:, P - 1:]
    lp_sec2 = torch.log_softmax(out2.logits.float()[:, :-1], -1).gather(2, ids[:, 1:, None]).squeeze(2)[:, P - 1:]

    win_o, lens_o = O.vit_window_index([grid], CFG)
    from transformers import vision_utils as V
    win_hf, cu_hf = V.get_vision_window_index(torch.tensor([grid]), 2, 56, 14)
    assert torch.equal(win_o, win_hf) and [0] + list(np.cumsum(lens_o)) == cu_hf.tolist()
    vit_o = O.vit_forward(w, CFG, rows, [grid])
    assert torch.allclose(vit_o, vit, atol=2e-5, rtol=1e-4), (vit_o - vit).abs().max()
    lg_o = O.full_logits(w, CFG, ids[0], rows, [grid])
    err = (lg_o - logits[0]).abs().max().item()
    assert err < 5e-5, err
    lp_o = O.completion_logps(w, CFG, prompt, comps, rows, [grid])
    err_lp = (lp_o - lp).abs().max().item()
    assert err_lp < 5e-5, err_lp
    pos_o, delta = O.mrope_position_ids(ids[0].tolist(), [grid], CFG)
    pos_hf, delta_hf = m.model.get_rope_index(ids[:1], mm[:1], video_grid_thw=torch.tensor([grid]))
    assert torch.equal(pos_o, pos_hf[:, 0]) and delta == int(delta_hf[0]), (pos_o, pos_hf)
    pos_o2, delta2 = O.mrope_position_ids(ids[0].tolist(), [grid], CFG, second_per_grid_ts=[2.0])
    pos_hf2, delta_hf2 = m.model.get_rope_index(ids[:1], mm[:1], video_grid_thw=torch.tensor([grid]), second_per_grid_ts=torch.tensor([2.0]))
    assert torch.equal(pos_o2, pos_hf2[:, 0]) and delta2 == int(delta_hf2[0]) and not torch.equal(pos_o2, pos_o)
    print(f"oracle vs HF (Qwen2.5-VL): vit max err {(vit_o - vit).abs().max():.2e}, logits {err:.2e}, logps {err_lp:.2e}")

    blob = {"w::" + k: v.numpy().astype(np.float16) for k, v in w.items()}
    for k, v in blob.items():
        assert np.array_equal(v.astype(np.float32), w[k[3:]].numpy()), k
    blob.update(
        cfg=np.frombuffer(json.dumps(CFG).encode(), dtype=np.uint8),
        frames=frames.numpy(), grid=np.array(grid), prompt=prompt.numpy(), completions=comps.numpy(),
        hf_vit=vit.numpy(), hf_logits_row0=logits[0].numpy(), hf_logps=lp.numpy(), hf_logps_sec2=lp_sec2.numpy(),
        hf_pos=pos_hf[:, 0].numpy(), hf_delta=np.array(int(delta_hf[0])), hf_pos_sec2=pos_hf2[:, 0].numpy(),
        hf_window_index=win_hf.numpy(), hf_cu_window=cu_hf.numpy(),
    )
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
