"""Architecture description of the Qwen2-VL / Qwen2.5-VL families the SG-RLVR trainer drives
(SG_RLVR_trainer.py:182-190 dispatches "Qwen2-VL" ids to Qwen2VLForConditionalGeneration and "Qwen2.5-VL" ids -- what
run_SpaceR_SG_RLVR.sh:16 trains -- to Qwen2_5_VLForConditionalGeneration).  The language model is the same decoder; the
vision towers differ (vit_kind)."""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import Tuple


@dataclass(frozen=True)
class Qwen2VLConfig:
    # language model
    hidden: int
    layers: int
    heads: int
    kv_heads: int
    intermediate: int
    vocab: int
    head_dim: int = 128
    mrope_section: Tuple[int, int, int] = (16, 24, 24)
    rope_theta: float = 1e6
    rms_eps: float = 1e-6
    tie_embeddings: bool = False
    # vision tower
    vit_dim: int = 1280
    vit_depth: int = 32
    vit_heads: int = 16
    vit_mlp: int = 5120
    patch: int = 14
    tpatch: int = 2
    merge: int = 2
    # "qwen2": LayerNorm + fc1/quick_gelu/fc2 blocks, per-frame attention everywhere.
    # "qwen2_5": RMSNorm + biased SwiGLU blocks, WINDOW attention (vit_window px) except on vit_fullatt blocks, RMSNorm
    # merger, temporal M-RoPE positions scaled by tokens_per_second * second_per_grid_t (HF modeling_qwen2_5_vl.py:345-466).
    vit_kind: str = "qwen2"
    vit_window: int = 112
    vit_fullatt: Tuple[int, ...] = ()
    tokens_per_second: int = 2
    # special ids (HF Qwen2VLConfig defaults)
    image_token_id: int = 151655
    video_token_id: int = 151656
    vision_start_id: int = 151652
    vision_end_id: int = 151653
    eos_token_id: int = 151645
    pad_token_id: int = 151643

    @property
    def vit_head_dim(self) -> int:
        return self.vit_dim // self.vit_heads

    @property
    def patch_k(self) -> int:
        return 3 * self.tpatch * self.patch * self.patch          # 1176

    @property
    def patch_kpad(self) -> int:
        return (self.patch_k + 63) // 64 * 64                      # 1216: GEMM contraction dim padded to 64

    @property
    def vit_mlp_pad(self) -> int:
        """qwen2_5: SwiGLU width padded to a multiple of 64 with zero rows/columns (3420 -> 3456) so that gate|up and
        down are plain GEMM shapes; silu(0) * 0 = 0 keeps the padding inert."""
        return (self.vit_mlp + 63) // 64 * 64

    @property
    def qkv_dim(self) -> int:
        return (self.heads + 2 * self.kv_heads) * self.head_dim

    def as_oracle_dict(self) -> dict:
        """The plain dict oracle/qwen2vl_fp32.py consumes (tests only)."""
        d = asdict(self)
        d["mrope_section"] = tuple(self.mrope_section)
        return d


QWEN2_VL_7B = Qwen2VLConfig(hidden=3584, layers=28, heads=28, kv_heads=4, intermediate=18944, vocab=152064)
QWEN2_VL_2B = Qwen2VLConfig(hidden=1536, layers=28, heads=12, kv_heads=2, intermediate=8960, vocab=151936,
                            tie_embeddings=True)
QWEN2_5_VL_7B = Qwen2VLConfig(hidden=3584, layers=28, heads=28, kv_heads=4, intermediate=18944, vocab=152064,
                              vit_kind="qwen2_5", vit_mlp=3420, vit_fullatt=(7, 15, 23, 31))
QWEN2_5_VL_3B = Qwen2VLConfig(hidden=2048, layers=36, heads=16, kv_heads=2, intermediate=11008, vocab=151936,
                              tie_embeddings=True, vit_kind="qwen2_5", vit_mlp=3420, vit_fullatt=(7, 15, 23, 31))
# small shapes that keep the real head sizes (LLM 128, ViT 80) -- the golden fixture's config
TINY = Qwen2VLConfig(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=1024, vit_dim=320,
                     vit_depth=2, vit_heads=4, vit_mlp=1280, image_token_id=1000, video_token_id=1001,
                     vision_start_id=1002, vision_end_id=1003, eos_token_id=7, pad_token_id=0)

# the same miniature with lm_head tied to the embedding table (the Qwen2-VL-2B arrangement); fixture tiny_tied_model.npz
TINY_TIED = Qwen2VLConfig(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=1024, vit_dim=320,
                          vit_depth=2, vit_heads=4, vit_mlp=1280, image_token_id=1000, video_token_id=1001,
                          vision_start_id=1002, vision_end_id=1003, eos_token_id=7, pad_token_id=0, tie_embeddings=True)

# Qwen2.5-VL in miniature: 4 vision blocks (1 and 3 full attention), 56-px windows = 2x2 merge units, ragged SwiGLU width
TINY25 = Qwen2VLConfig(hidden=256, layers=2, heads=2, kv_heads=1, intermediate=512, vocab=1024, vit_dim=320,
                       vit_depth=4, vit_heads=4, vit_mlp=420, image_token_id=1000, video_token_id=1001,
                       vision_start_id=1002, vision_end_id=1003, eos_token_id=7, pad_token_id=0,
                       vit_kind="qwen2_5", vit_window=56, vit_fullatt=(1, 3))

PRESETS = {"Qwen2.5-VL-7B": QWEN2_5_VL_7B, "Qwen2.5-VL-3B": QWEN2_5_VL_3B, "Qwen2-VL-7B": QWEN2_VL_7B, "Qwen2-VL-2B": QWEN2_VL_2B,
           "tiny25": TINY25, "tiny_tied": TINY_TIED, "tiny": TINY}


def preset_for(model_id: str) -> Qwen2VLConfig:
    for k, v in PRESETS.items():
        if k.lower() in model_id.lower():
            return v
    raise KeyError(f"no Qwen2-VL preset matches {model_id!r}")
