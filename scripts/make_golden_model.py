"""Generate tests/golden/tiny_model.npz by running HF transformers' Qwen2VLForConditionalGeneration
(the third-party model the reference trainer calls: SG_RLVR_trainer.py:183,357,463) in THIS
container, random-init, fp32, eager attention, on a tiny config that keeps the real head sizes
(LLM head_dim 128, ViT head_dim 80).  Run once by hand:

    python scripts/make_golden_model.py

It also asserts that oracle/qwen2vl_fp32.py reproduces HF to fp32 round-off, which is what
"pins" the oracle (the reference itself has no golden vectors for the model arithmetic).
The fixture holds: the config, the weights (fp16-representable values stored as fp16 to stay
small), the synthetic inputs and HF's outputs.  Neither HF nor /root/reference is needed to
consume it.
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import qwen2vl_fp32 as O  # noqa: E402

# ``--tied`` writes tiny_tied_model.npz: the same miniature with tie_word_embeddings=True, the Qwen2-VL-2B arrangement
# (lm_head IS the embedding table: BASELINE.json configs[0]/[1] are 2B models)
TIED = "--tied" in sys.argv
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "tiny_tied_model.npz" if TIED else "tiny_model.npz")

CFG = O.make_config(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=1024,
                    vit_dim=320, vit_depth=2, vit_heads=4, vit_mlp=1280, head_dim=128,
                    video_token_id=1001, image_token_id=1000, tie_embeddings=TIED)
VISION_START, VISION_END = 1002, 1003


def hf_model():
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                         intermediate_size=512, vocab_size=1024, rms_norm_eps=1e-6,
                         rope_parameters=dict(rope_theta=1e6, rope_type="default", mrope_section=[16, 24, 24]),
                         max_position_embeddings=4096, tie_word_embeddings=TIED),
        vision_config=dict(depth=2, embed_dim=320, hidden_size=256, num_heads=4, mlp_ratio=4, patch_size=14,
                           spatial_merge_size=2, temporal_patch_size=2, in_channels=3),
        image_token_id=1000, video_token_id=1001, vision_start_token_id=1002, vision_end_token_id=1003,
        tie_word_embeddings=TIED,
    )
    cfg._attn_implementation = "eager"
    torch.manual_seed(8 if TIED else 7)
    m = Qwen2VLForConditionalGeneration(cfg).float().eval()
    if TIED:
        assert m.lm_head.weight.data_ptr() == m.model.language_model.embed_tokens.weight.data_ptr(), "HF did not tie the weights"
    # HF init leaves biases at zero / norms at one; perturb so every term is exercised.
    g = torch.Generator().manual_seed(11)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 1:
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
            # keep values fp16-representable so the fixture can be stored compactly and exactly
            p.copy_(p.half().float())
    return m


def to_ckpt_names(sd):
    """transformers 5.x state_dict names -> original Qwen2-VL checkpoint names."""
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
    if TIED:
        out.pop("lm_head.weight", None)          # tied checkpoints carry the embedding table only
    return out


def main():
    m = hf_model()
    w = to_ckpt_names(m.state_dict())
    g = torch.Generator().manual_seed(3)
    frames = torch.randint(0, 256, (4, 3, 56, 84), generator=g, dtype=torch.uint8)   # grid (2, 4, 6)
    rows, grid = O.patchify_frames(frames, CFG)
    nv = grid[0] * grid[1] * grid[2] // 4
    text = torch.randint(5, 990, (9,), generator=g)
    prompt = torch.cat([torch.tensor([VISION_START]), torch.full((nv,), 1001), torch.tensor([VISION_END]), text])
    comps = torch.randint(5, 990, (3, 6), generator=g)
    P = prompt.numel()

    # --- HF run: K independent rows prompt+completion, the shape the reference's logps call sees
    ids = torch.stack([torch.cat([prompt, c]) for c in comps])
    K = ids.shape[0]
    mm = ((ids == 1001) * 2 + (ids == 1000) * 1).int()
    with torch.no_grad():
        out = m(input_ids=ids, pixel_values_videos=rows.repeat(K, 1),
                video_grid_thw=torch.tensor([grid] * K), mm_token_type_ids=mm)
        vit = m.model.visual(rows, grid_thw=torch.tensor([grid])).pooler_output
    logits = out.logits.float()
    lp = torch.log_softmax(logits[:, :-1], -1).gather(2, ids[:, 1:, None]).squeeze(2)[:, P - 1:]

    # --- oracle must agree
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
    print(f"oracle vs HF: vit max err {(vit_o - vit).abs().max():.2e}, logits {err:.2e}, logps {err_lp:.2e}")

    blob = {"w::" + k: v.numpy().astype(np.float16) for k, v in w.items()}
    for k, v in blob.items():   # exactness of the fp16 storage
        assert np.array_equal(v.astype(np.float32), w[k[3:]].numpy()), k
    blob.update(
        cfg=np.frombuffer(json.dumps(CFG).encode(), dtype=np.uint8),
        frames=frames.numpy(), grid=np.array(grid), prompt=prompt.numpy(), completions=comps.numpy(),
        hf_vit=vit.numpy(), hf_logits_row0=logits[0].numpy(), hf_logps=lp.numpy(),
        hf_pos=pos_hf[:, 0].numpy(), hf_delta=np.array(int(delta_hf[0])),
    )
    np.savez_compressed(OUT, **blob)
    print("wrote", OUT, os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
