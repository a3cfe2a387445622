"""Parameter storage for the engine: ONE flat bf16 buffer (what the kernels read) with named 2-D/1-D views,
laid out so that fused GEMMs see contiguous operands (q|k|v and gate|up are concatenated along the output
dim, the patch-embed contraction dim is zero-padded 1176 -> 1216).  The trainer attaches a flat fp32 master
copy, flat fp32 gradients and Adam moments with the same offsets, so the optimizer and the data-parallel
all-reduce each touch a single contiguous range (288 GB of HBM: replicas instead of ZeRO-3).

Checkpoint names are the original Qwen2-VL ones (``visual.*``, ``model.layers.*``, ``lm_head.weight``) so a
safetensors state dict of the model the reference trains (SG_RLVR_trainer.py:183) maps 1:1.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

import torch

from .config import Qwen2VLConfig

ALIGN = 64  # elements; keeps every view 128-byte aligned in bf16 and 16-byte aligned for vector loads


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    offset: int = 0

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def param_specs(cfg: Qwen2VLConfig) -> List[ParamSpec]:
    D, H, I, V = cfg.vit_dim, cfg.hidden, cfg.intermediate, cfg.vocab
    m4 = cfg.merge ** 2 * D
    sp: List[ParamSpec] = [ParamSpec("vit.patch_w", (D, cfg.patch_kpad))]
    v25 = cfg.vit_kind == "qwen2_5"
    Ip = cfg.vit_mlp_pad
    for i in range(cfg.vit_depth):
        p = f"vit.{i}."
        if v25:     # RMSNorm blocks, gate|up fused and zero-padded to Ip rows each, down zero-padded to Ip columns
            sp += [ParamSpec(p + "n1_w", (D,)), ParamSpec(p + "qkv_w", (3 * D, D)), ParamSpec(p + "qkv_b", (3 * D,)),
                   ParamSpec(p + "proj_w", (D, D)), ParamSpec(p + "proj_b", (D,)), ParamSpec(p + "n2_w", (D,)),
                   ParamSpec(p + "gu_w", (2 * Ip, D)), ParamSpec(p + "gu_b", (2 * Ip,)),
                   ParamSpec(p + "down_w", (D, Ip)), ParamSpec(p + "down_b", (D,))]
        else:
            sp += [ParamSpec(p + "n1_w", (D,)), ParamSpec(p + "n1_b", (D,)),
                   ParamSpec(p + "qkv_w", (3 * D, D)), ParamSpec(p + "qkv_b", (3 * D,)),
                   ParamSpec(p + "proj_w", (D, D)), ParamSpec(p + "proj_b", (D,)),
                   ParamSpec(p + "n2_w", (D,)), ParamSpec(p + "n2_b", (D,)),
                   ParamSpec(p + "fc1_w", (cfg.vit_mlp, D)), ParamSpec(p + "fc1_b", (cfg.vit_mlp,)),
                   ParamSpec(p + "fc2_w", (D, cfg.vit_mlp)), ParamSpec(p + "fc2_b", (D,))]
    sp += [ParamSpec("merger.ln_w", (D,))] + ([] if v25 else [ParamSpec("merger.ln_b", (D,))])
    sp += [ParamSpec("merger.m0_w", (m4, m4)), ParamSpec("merger.m0_b", (m4,)),
           ParamSpec("merger.m2_w", (H, m4)), ParamSpec("merger.m2_b", (H,))]
    sp.append(ParamSpec("llm.embed", (V, H)))
    for i in range(cfg.layers):
        p = f"llm.{i}."
        sp += [ParamSpec(p + "ln1_w", (H,)), ParamSpec(p + "qkv_w", (cfg.qkv_dim, H)), ParamSpec(p + "qkv_b", (cfg.qkv_dim,)),
               ParamSpec(p + "o_w", (H, cfg.heads * cfg.head_dim)), ParamSpec(p + "ln2_w", (H,)),
               ParamSpec(p + "gu_w", (2 * I, H)), ParamSpec(p + "down_w", (H, I))]
    sp.append(ParamSpec("llm.norm_w", (H,)))
    if not cfg.tie_embeddings:
        sp.append(ParamSpec("llm.lm_head", (V, H)))
    off = 0
    for s in sp:
        s.offset = off
        off += (s.numel + ALIGN - 1) // ALIGN * ALIGN
    return sp


def total_numel(specs: Iterable[ParamSpec]) -> int:
    last = list(specs)[-1]
    return last.offset + (last.numel + ALIGN - 1) // ALIGN * ALIGN


class FlatParams:
    """A flat tensor + dict of named views (no copies)."""

    def __init__(self, cfg: Qwen2VLConfig, flat: torch.Tensor, specs: List[ParamSpec] = None):
        self.cfg = cfg
        self.specs = specs or param_specs(cfg)
        assert flat.numel() == total_numel(self.specs)
        self.flat = flat
        self.v: Dict[str, torch.Tensor] = {s.name: flat[s.offset:s.offset + s.numel].view(*s.shape) for s in self.specs}
        if cfg.tie_embeddings:
            self.v["llm.lm_head"] = self.v["llm.embed"]

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.v[name]

    @classmethod
    def empty(cls, cfg, device, dtype=torch.bfloat16):
        specs = param_specs(cfg)
        return cls(cfg, torch.zeros(total_numel(specs), device=device, dtype=dtype), specs)

    def like(self, dtype) -> "FlatParams":
        return FlatParams(self.cfg, torch.zeros(self.flat.numel(), device=self.flat.device, dtype=dtype), self.specs)


# ------------------------------------------------------------------------------------ name mapping
def _ckpt_to_engine(cfg: Qwen2VLConfig, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Original Qwen2-VL checkpoint names -> engine (fused) tensors, still on the source device/dtype."""
    out: Dict[str, torch.Tensor] = {}
    pw = sd["visual.patch_embed.proj.weight"].reshape(cfg.vit_dim, -1)
    pad = torch.zeros(cfg.vit_dim, cfg.patch_kpad, dtype=pw.dtype)
    pad[:, :cfg.patch_k] = pw
    out["vit.patch_w"] = pad
    v25 = cfg.vit_kind == "qwen2_5"
    I, Ip = cfg.vit_mlp, cfg.vit_mlp_pad
    for i in range(cfg.vit_depth):
        s, d = f"visual.blocks.{i}.", f"vit.{i}."
        if v25:
            for a, b in (("norm1.weight", "n1_w"), ("attn.qkv.weight", "qkv_w"), ("attn.qkv.bias", "qkv_b"),
                         ("attn.proj.weight", "proj_w"), ("attn.proj.bias", "proj_b"), ("norm2.weight", "n2_w"),
                         ("mlp.down_proj.bias", "down_b")):
                out[d + b] = sd[s + a]
            g, u = sd[s + "mlp.gate_proj.weight"], sd[s + "mlp.up_proj.weight"]
            gu = torch.zeros(2 * Ip, cfg.vit_dim, dtype=g.dtype)
            gu[:I] = g; gu[Ip:Ip + I] = u
            gb = torch.zeros(2 * Ip, dtype=g.dtype)
            gb[:I] = sd[s + "mlp.gate_proj.bias"]; gb[Ip:Ip + I] = sd[s + "mlp.up_proj.bias"]
            dw = torch.zeros(cfg.vit_dim, Ip, dtype=g.dtype)
            dw[:, :I] = sd[s + "mlp.down_proj.weight"]
            out[d + "gu_w"], out[d + "gu_b"], out[d + "down_w"] = gu, gb, dw
        else:
            for a, b in (("norm1.weight", "n1_w"), ("norm1.bias", "n1_b"), ("attn.qkv.weight", "qkv_w"), ("attn.qkv.bias", "qkv_b"),
                         ("attn.proj.weight", "proj_w"), ("attn.proj.bias", "proj_b"), ("norm2.weight", "n2_w"),
                         ("norm2.bias", "n2_b"), ("mlp.fc1.weight", "fc1_w"), ("mlp.fc1.bias", "fc1_b"),
                         ("mlp.fc2.weight", "fc2_w"), ("mlp.fc2.bias", "fc2_b")):
                out[d + b] = sd[s + a]
    for a, b in (("ln_q.weight", "ln_w"), ("ln_q.bias", "ln_b"), ("mlp.0.weight", "m0_w"), ("mlp.0.bias", "m0_b"),
                 ("mlp.2.weight", "m2_w"), ("mlp.2.bias", "m2_b")):
        if not (v25 and b == "ln_b"):
            out["merger." + b] = sd["visual.merger." + a]
    out["llm.embed"] = sd["model.embed_tokens.weight"]
    for i in range(cfg.layers):
        s, d = f"model.layers.{i}.", f"llm.{i}."
        out[d + "ln1_w"] = sd[s + "input_layernorm.weight"]
        out[d + "qkv_w"] = torch.cat([sd[s + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        out[d + "qkv_b"] = torch.cat([sd[s + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        out[d + "o_w"] = sd[s + "self_attn.o_proj.weight"]
        out[d + "ln2_w"] = sd[s + "post_attention_layernorm.weight"]
        out[d + "gu_w"] = torch.cat([sd[s + "mlp.gate_proj.weight"], sd[s + "mlp.up_proj.weight"]], 0)
        out[d + "down_w"] = sd[s + "mlp.down_proj.weight"]
    out["llm.norm_w"] = sd["model.norm.weight"]
    if not cfg.tie_embeddings:
        out["llm.lm_head"] = sd["lm_head.weight"]
    return out


def load_state_dict(params: FlatParams, sd: Dict[str, torch.Tensor]) -> None:
    eng = _ckpt_to_engine(params.cfg, sd)
    for s in params.specs:
        params[s.name].copy_(eng[s.name].to(params.flat.dtype))


def export_state_dict(params: FlatParams) -> Dict[str, torch.Tensor]:
    """Engine tensors -> original checkpoint names (for save_model / SpaceR-Eval consumption)."""
    cfg, v = params.cfg, params.v
    sd: Dict[str, torch.Tensor] = {}
    sd["visual.patch_embed.proj.weight"] = v["vit.patch_w"][:, :cfg.patch_k].reshape(
        cfg.vit_dim, 3, cfg.tpatch, cfg.patch, cfg.patch).clone()
    v25 = cfg.vit_kind == "qwen2_5"
    I, Ip = cfg.vit_mlp, cfg.vit_mlp_pad
    for i in range(cfg.vit_depth):
        s, d = f"visual.blocks.{i}.", f"vit.{i}."
        if v25:
            for a, b in (("norm1.weight", "n1_w"), ("attn.qkv.weight", "qkv_w"), ("attn.qkv.bias", "qkv_b"),
                         ("attn.proj.weight", "proj_w"), ("attn.proj.bias", "proj_b"), ("norm2.weight", "n2_w"),
                         ("mlp.down_proj.bias", "down_b")):
                sd[s + a] = v[d + b].clone()
            sd[s + "mlp.gate_proj.weight"] = v[d + "gu_w"][:I].clone()
            sd[s + "mlp.up_proj.weight"] = v[d + "gu_w"][Ip:Ip + I].clone()
            sd[s + "mlp.gate_proj.bias"] = v[d + "gu_b"][:I].clone()
            sd[s + "mlp.up_proj.bias"] = v[d + "gu_b"][Ip:Ip + I].clone()
            sd[s + "mlp.down_proj.weight"] = v[d + "down_w"][:, :I].clone()
        else:
            for a, b in (("norm1.weight", "n1_w"), ("norm1.bias", "n1_b"), ("attn.qkv.weight", "qkv_w"), ("attn.qkv.bias", "qkv_b"),
                         ("attn.proj.weight", "proj_w"), ("attn.proj.bias", "proj_b"), ("norm2.weight", "n2_w"),
                         ("norm2.bias", "n2_b"), ("mlp.fc1.weight", "fc1_w"), ("mlp.fc1.bias", "fc1_b"),
                         ("mlp.fc2.weight", "fc2_w"), ("mlp.fc2.bias", "fc2_b")):
                sd[s + a] = v[d + b].clone()
    for a, b in (("ln_q.weight", "ln_w"), ("ln_q.bias", "ln_b"), ("mlp.0.weight", "m0_w"), ("mlp.0.bias", "m0_b"),
                 ("mlp.2.weight", "m2_w"), ("mlp.2.bias", "m2_b")):
        if not (v25 and b == "ln_b"):
            sd["visual.merger." + a] = v["merger." + b].clone()
    sd["model.embed_tokens.weight"] = v["llm.embed"].clone()
    qd, kd = cfg.heads * cfg.head_dim, cfg.kv_heads * cfg.head_dim
    for i in range(cfg.layers):
        s, d = f"model.layers.{i}.", f"llm.{i}."
        sd[s + "input_layernorm.weight"] = v[d + "ln1_w"].clone()
        for n, (a, b) in zip("qkv", ((0, qd), (qd, qd + kd), (qd + kd, qd + 2 * kd))):
            sd[s + f"self_attn.{n}_proj.weight"] = v[d + "qkv_w"][a:b].clone()
            sd[s + f"self_attn.{n}_proj.bias"] = v[d + "qkv_b"][a:b].clone()
        sd[s + "self_attn.o_proj.weight"] = v[d + "o_w"].clone()
        sd[s + "post_attention_layernorm.weight"] = v[d + "ln2_w"].clone()
        sd[s + "mlp.gate_proj.weight"] = v[d + "gu_w"][:cfg.intermediate].clone()
        sd[s + "mlp.up_proj.weight"] = v[d + "gu_w"][cfg.intermediate:].clone()
        sd[s + "mlp.down_proj.weight"] = v[d + "down_w"].clone()
    sd["model.norm.weight"] = v["llm.norm_w"].clone()
    if not cfg.tie_embeddings:
        sd["lm_head.weight"] = v["llm.lm_head"].clone()
    return sd


def random_init_(params: FlatParams, seed: int = 1234, std: float = 0.02) -> None:
    """Synthetic weights of BASELINE.md: N(0, 0.02), norm weights 1, norm biases 0 (generated on-device)."""
    g = torch.Generator(device=params.flat.device).manual_seed(seed)
    for s in params.specs:
        v = params[s.name]
        base = s.name.rsplit(".", 1)[-1]
        if base in ("n1_w", "n2_w", "ln_w", "ln1_w", "ln2_w", "norm_w"):
            v.fill_(1.0)
        elif base in ("n1_b", "n2_b", "ln_b"):
            v.zero_()
        else:
            chunk = 1 << 26
            flat = v.view(-1)
            for a in range(0, flat.numel(), chunk):
                b = min(flat.numel(), a + chunk)
                flat[a:b] = (torch.randn(b - a, device=flat.device, generator=g) * std).to(flat.dtype)
    if "vit.patch_w" in params.v:
        params["vit.patch_w"][:, params.cfg.patch_k:] = 0
    cfg = params.cfg
    if cfg.vit_kind == "qwen2_5" and cfg.vit_mlp_pad != cfg.vit_mlp:       # the SwiGLU padding stays exactly zero
        I, Ip = cfg.vit_mlp, cfg.vit_mlp_pad
        for i in range(cfg.vit_depth):
            p = f"vit.{i}."
            params[p + "gu_w"][I:Ip] = 0; params[p + "gu_w"][Ip + I:] = 0
            params[p + "gu_b"][I:Ip] = 0; params[p + "gu_b"][Ip + I:] = 0
            params[p + "down_w"][:, I:] = 0
