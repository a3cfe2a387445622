"""Checkpoints in the layout the reference's ecosystem reads and writes.

``trainer.save_model(output_dir)`` in the reference (open_r1/SG-RLVR.py:377-384) goes through HF ``Trainer.save_model``:
safetensors in the original Qwen2-VL / Qwen2.5-VL tensor names, ``config.json``, and the tokenizer / processor files, so
that SpaceR-Eval can ``from_pretrained`` the directory (EV/data_utils/vsibench.py:79-93).  This module writes the same
artefact from the engine's flat buffers and reads it (or an original HF snapshot) back:

  * tensors: `weights.export_state_dict` names (``visual.*``, ``model.layers.*``, ``lm_head.weight``), bf16;
  * ``config.json``: the source snapshot's file when the model was loaded from a directory, else the era-style (flat
    text fields + ``vision_config``) dict below -- transformers 4.49 .. 5.15 load either;
  * tokenizer / processor / generation files: copied from the source snapshot and/or ``processor.save_pretrained``.
tests/test_checkpoint_hf.py loads the written directory with HF transformers and compares logits with the golden vectors.
"""
from __future__ import annotations

import json
import os
import shutil
from typing import Dict, Optional

import torch

from .config import Qwen2VLConfig
from .weights import FlatParams, export_state_dict

PASSTHROUGH = ("generation_config.json", "tokenizer.json", "tokenizer_config.json", "vocab.json", "merges.txt",
               "special_tokens_map.json", "added_tokens.json", "preprocessor_config.json", "video_preprocessor_config.json",
               "chat_template.json", "chat_template.jinja")


def hf_config_dict(cfg: Qwen2VLConfig) -> dict:
    """``config.json`` of a Qwen2-VL / Qwen2.5-VL checkpoint for this architecture (era layout)."""
    v25 = cfg.vit_kind == "qwen2_5"
    d = dict(architectures=["Qwen2_5_VLForConditionalGeneration" if v25 else "Qwen2VLForConditionalGeneration"],
             model_type="qwen2_5_vl" if v25 else "qwen2_vl",
             hidden_size=cfg.hidden, intermediate_size=cfg.intermediate, num_hidden_layers=cfg.layers,
             num_attention_heads=cfg.heads, num_key_value_heads=cfg.kv_heads, vocab_size=cfg.vocab, rms_norm_eps=cfg.rms_eps,
             rope_theta=cfg.rope_theta, rope_scaling=dict(type="mrope", mrope_section=list(cfg.mrope_section)),
             max_position_embeddings=32768, hidden_act="silu", tie_word_embeddings=cfg.tie_embeddings, torch_dtype="bfloat16",
             image_token_id=cfg.image_token_id, video_token_id=cfg.video_token_id, vision_start_token_id=cfg.vision_start_id,
             vision_end_token_id=cfg.vision_end_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.pad_token_id)
    if v25:
        d["vision_config"] = dict(depth=cfg.vit_depth, hidden_size=cfg.vit_dim, out_hidden_size=cfg.hidden, num_heads=cfg.vit_heads,
                                  intermediate_size=cfg.vit_mlp, patch_size=cfg.patch, spatial_merge_size=cfg.merge,
                                  temporal_patch_size=cfg.tpatch, in_chans=3, window_size=cfg.vit_window,
                                  fullatt_block_indexes=list(cfg.vit_fullatt), tokens_per_second=cfg.tokens_per_second,
                                  hidden_act="silu")
    else:
        d["vision_config"] = dict(depth=cfg.vit_depth, embed_dim=cfg.vit_dim, hidden_size=cfg.hidden, num_heads=cfg.vit_heads,
                                  mlp_ratio=cfg.vit_mlp // cfg.vit_dim, patch_size=cfg.patch, spatial_merge_size=cfg.merge,
                                  temporal_patch_size=cfg.tpatch, in_chans=3)
    return d


def config_from_hf(d: dict) -> Qwen2VLConfig:
    """The engine's architecture description from a HF ``config.json`` (era layout or the 5.x ``text_config`` nesting)."""
    t = dict(d.get("text_config") or {})
    for k, v in d.items():                                   # era layout keeps the text fields at the top level
        t.setdefault(k, v)
    v = d["vision_config"]
    rope = t.get("rope_scaling") or t.get("rope_parameters") or {}
    v25 = d.get("model_type", "").startswith("qwen2_5") or "fullatt_block_indexes" in v
    heads = t["num_attention_heads"]
    kw = dict(hidden=t["hidden_size"], layers=t["num_hidden_layers"], heads=heads, kv_heads=t["num_key_value_heads"],
              intermediate=t["intermediate_size"], vocab=t["vocab_size"], head_dim=t.get("head_dim") or t["hidden_size"] // heads,
              mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))),
              rope_theta=float(rope.get("rope_theta", t.get("rope_theta", 1e6))), rms_eps=float(t.get("rms_norm_eps", 1e-6)),
              tie_embeddings=bool(d.get("tie_word_embeddings", t.get("tie_word_embeddings", False))),
              vit_depth=v["depth"], vit_heads=v["num_heads"], patch=v.get("patch_size", 14), tpatch=v.get("temporal_patch_size", 2),
              merge=v.get("spatial_merge_size", 2))
    for ours, theirs in (("image_token_id", "image_token_id"), ("video_token_id", "video_token_id"),
                         ("vision_start_id", "vision_start_token_id"), ("vision_end_id", "vision_end_token_id"),
                         ("eos_token_id", "eos_token_id"), ("pad_token_id", "pad_token_id")):
        val = d.get(theirs, t.get(theirs))
        if isinstance(val, (list, tuple)):
            val = val[0]
        if val is not None:
            kw[ours] = int(val)
    if v25:
        kw.update(vit_kind="qwen2_5", vit_dim=v["hidden_size"], vit_mlp=v["intermediate_size"], vit_window=v.get("window_size", 112),
                  vit_fullatt=tuple(v.get("fullatt_block_indexes", ())), tokens_per_second=int(v.get("tokens_per_second", 2)))
    else:
        kw.update(vit_dim=v["embed_dim"], vit_mlp=int(v["embed_dim"] * v.get("mlp_ratio", 4)))
    return Qwen2VLConfig(**kw)


def config_of_dir(path: str) -> Optional[Qwen2VLConfig]:
    f = os.path.join(path, "config.json")
    if os.path.isdir(path) and os.path.exists(f):
        with open(f) as fh:
            return config_from_hf(json.load(fh))
    return None


def write_checkpoint(output_dir: str, params: FlatParams, *, source_dir: Optional[str] = None, processor=None,
                     extra_state: Optional[dict] = None) -> None:
    os.makedirs(output_dir, exist_ok=True)
    sd = {k: v.detach().to("cpu", torch.bfloat16).contiguous() for k, v in export_state_dict(params).items()}
    try:
        from safetensors.torch import save_file
        save_file(sd, os.path.join(output_dir, "model.safetensors"), metadata={"format": "pt"})
    except ImportError:
        torch.save(sd, os.path.join(output_dir, "pytorch_model.bin"))
    src_cfg = os.path.join(source_dir, "config.json") if source_dir and os.path.isdir(source_dir) else None
    if src_cfg and os.path.exists(src_cfg):
        shutil.copyfile(src_cfg, os.path.join(output_dir, "config.json"))
    else:
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(hf_config_dict(params.cfg), f, indent=1)
    if source_dir and os.path.isdir(source_dir):
        for name in PASSTHROUGH:
            src = os.path.join(source_dir, name)
            if os.path.exists(src):
                shutil.copyfile(src, os.path.join(output_dir, name))
    if processor is not None and hasattr(processor, "save_pretrained"):
        processor.save_pretrained(output_dir)
    if extra_state is not None:
        with open(os.path.join(output_dir, "trainer_state.json"), "w") as f:
            json.dump(extra_state, f)


def _weight_files(path: str):
    """The weight shards of an HF-layout directory: the files the ``*.index.json`` names when there is one, else
    ``model*.safetensors``, else ``pytorch_model*.bin`` -- safetensors preferred when both exist; ``training_args.bin``,
    ``optimizer.pt``, ``scheduler.pt``, ``rng_state*.pth`` (HF Trainer outputs) are never weight files."""
    names = sorted(os.listdir(path))
    for index in ("model.safetensors.index.json", "pytorch_model.bin.index.json"):
        if index in names:
            with open(os.path.join(path, index)) as f:
                return sorted({os.path.join(path, v) for v in json.load(f)["weight_map"].values()})
    st = [n for n in names if n.endswith(".safetensors") and (n.startswith("model") or n == "adapter_model.safetensors")]
    if not st:
        st = [n for n in names if n.endswith(".safetensors")]
    if st:
        return [os.path.join(path, n) for n in st]
    return [os.path.join(path, n) for n in names if n.startswith("pytorch_model") and n.endswith(".bin")]


def read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    """A directory (or file) of safetensors / .bin shards in the original names (transformers 5.x prefixes accepted).  A hub id
    such as the launch script's ``Qwen/Qwen2.5-VL-7B-Instruct`` is resolved through the local huggingface_hub cache."""
    if not os.path.exists(path):
        try:
            from huggingface_hub import snapshot_download
            path = snapshot_download(path, local_files_only=bool(os.environ.get("HF_HUB_OFFLINE", "")),
                                     allow_patterns=["*.safetensors", "*.json", "*.bin"])
        except Exception as e:           # noqa: BLE001
            raise FileNotFoundError(f"{path!r} is neither a local checkpoint directory nor a hub snapshot that could be "
                                    f"resolved ({type(e).__name__}: {e})") from e
    files = [path] if os.path.isfile(path) else _weight_files(path)
    if not files:
        raise FileNotFoundError(f"no checkpoint shards under {path}")
    sd: Dict[str, torch.Tensor] = {}
    for f in files:
        if f.endswith(".safetensors"):
            from safetensors.torch import load_file
            part = load_file(f)
        else:
            part = torch.load(f, map_location="cpu", weights_only=True)
        for k, v in part.items():
            k = k.replace("model.language_model.", "model.").replace("model.visual.", "visual.")
            sd[k] = v
    return sd
