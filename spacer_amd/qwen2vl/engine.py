"""Qwen2-VL forward / backward expressed as a sequence of libspacer_hip.so launches.

This replaces what the reference obtains from HF ``Qwen2VLForConditionalGeneration`` + autograd + flash-attn
(SG_RLVR_trainer.py:357 ``model(input_ids, **kwargs).logits`` and ``loss.backward()``).  Python only orders
the launches and owns the buffers; every tensor op is a hand-written gfx950 kernel.

Conventions
  * residual stream fp32, GEMM operands / activations bf16, gradients of parameters fp32 (accumulated in place)
  * one token-packed batch per call; attention structure given as segments (see include/spacer_hip.h)
  * no recompute: 288 GB of HBM holds every layer's activations for a 5.5k-token prompt group (~27 GB)
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from .. import kernels as K
from . import positions as POS
from .config import Qwen2VLConfig
from .weights import FlatParams

BF16, F32 = torch.bfloat16, torch.float32


class Qwen2VLEngine:
    HEAD_CHUNK = 8192      # vocabulary rows of lm_head per head chunk (fp32 chunk logits: 268 MB at 8192 completion rows)

    def __init__(self, cfg: Qwen2VLConfig, params: FlatParams, recompute: bool = False):
        self.cfg = cfg
        self.W = params
        self.dev = params.flat.device
        # activation recompute (the reference trains with --gradient_checkpointing true, run_SpaceR_SG_RLVR.sh:28; here a
        # SELECTIVE policy): the MLP's wide intermediates (gate|up, act: 2/3 of a decoder layer's saved bytes; fc1 / act of the
        # vision blocks) and the lm_head logits are not kept for the backward pass but recomputed from the saved GEMM input
        # by the SAME kernel launch as in the forward -- bit-identical gradients (tests/test_recompute_gpu.py), one more
        # gate|up GEMM per layer and one more lm_head GEMM.  17.5 + 2.5 of the 27 + 2.5 GB per 5.5k-token 7B group.
        self.recompute = recompute
        # bumped by whoever rewrites the weights (GRPOEngine.optimizer_step): a prefill tape kept by the rollout engine is only
        # reused by a scoring pass of the same weights
        self.weights_version = 0

    # ------------------------------------------------------------------ helpers
    def _dx(self, dy: torch.Tensor, name: str) -> torch.Tensor:
        """dX[T, in] = dY[T, out] . W[out, in]: contraction over W's ROW index, read in place (trans_b)."""
        return K.gemm(dy, self.W[name], trans_b=True)

    @staticmethod
    def _dw(gw: torch.Tensor, dy: torch.Tensor, x: torch.Tensor) -> None:
        """gw[N,K] (fp32) += dy[T,N]^T @ x[T,K]: both operands contraction-major, read in place (trans_a, trans_b)."""
        K.gemm(dy, x, trans_a=True, trans_b=True, out=gw, residual=gw)

    def _zeros(self, *shape, dtype=F32):
        return K.zeros(*shape, device=self.dev, dtype=dtype)

    def _empty(self, *shape, dtype=F32):
        return torch.empty(*shape, device=self.dev, dtype=dtype)

    # ================================================================== vision tower
    def vit_forward(self, pix: torch.Tensor, grids: Sequence[Tuple[int, int, int]], tape: Optional[dict] = None):
        """pix bf16 [Np, patch_kpad] (spacer_patchify output) -> merged video embeds bf16 [Np/4, hidden]."""
        if self.cfg.vit_kind == "qwen2_5":
            return self._vit25_forward(pix, grids, tape)
        cfg, W = self.cfg, self.W
        D, Hh, hd = cfg.vit_dim, cfg.vit_heads, cfg.vit_head_dim
        Np = pix.shape[0]
        cos, sin = POS.vit_tables(grids, cfg, self.dev)
        seg_list = POS.vit_segments(grids)
        segs = K.make_segments(seg_list, self.dev)
        max_q = max(s[1] for s in seg_list)
        scale = hd ** -0.5
        x = K.gemm_nt(pix, W["vit.patch_w"], out_dtype=F32)
        blocks: List[dict] = []
        for i in range(cfg.vit_depth):
            p = f"vit.{i}."
            mean1, rstd1 = self._empty(Np), self._empty(Np)
            h = K.layernorm_fwd(x, W[p + "n1_w"], W[p + "n1_b"], 1e-6, mean=mean1, rstd=rstd1)
            qkv = K.gemm_nt(h, W[p + "qkv_w"], bias=W[p + "qkv_b"])
            K.rope_(qkv, cos, sin, 2 * Hh, hd)
            o, lse = K.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], segs, max_q, Hh, Hh, hd, False, scale)
            x_mid = K.gemm_nt(o, W[p + "proj_w"], bias=W[p + "proj_b"], residual=x, out_dtype=F32)
            mean2, rstd2 = self._empty(Np), self._empty(Np)
            h2 = K.layernorm_fwd(x_mid, W[p + "n2_w"], W[p + "n2_b"], 1e-6, mean=mean2, rstd=rstd2)
            f1 = K.gemm_nt(h2, W[p + "fc1_w"], bias=W[p + "fc1_b"])
            a = K.act_fwd(f1, K.SPACER_ACT_QUICK_GELU)
            x_out = K.gemm_nt(a, W[p + "fc2_w"], bias=W[p + "fc2_b"], residual=x_mid, out_dtype=F32)
            if tape is not None:
                keep = not self.recompute
                blocks.append(dict(x_in=x, mean1=mean1, rstd1=rstd1, h=h, qkv=qkv, o=o, lse=lse, x_mid=x_mid, mean2=mean2,
                                   rstd2=rstd2, h2=h2, f1=f1 if keep else None, a=a if keep else None))
            x = x_out
        mean, rstd = self._empty(Np), self._empty(Np)
        hm = K.layernorm_fwd(x, W["merger.ln_w"], W["merger.ln_b"], 1e-6, mean=mean, rstd=rstd)
        m4 = cfg.merge ** 2
        hm4 = hm.view(Np // m4, m4 * D)
        m1 = K.gemm_nt(hm4, W["merger.m0_w"], bias=W["merger.m0_b"])
        g = K.act_fwd(m1, K.SPACER_ACT_GELU_ERF)
        out = K.gemm_nt(g, W["merger.m2_w"], bias=W["merger.m2_b"])
        if tape is not None:
            tape.update(pix=pix, blocks=blocks, x_last=x, mean=mean, rstd=rstd, hm4=hm4, m1=m1, g=g, cos=cos, sin=sin,
                        segs=segs, max_q=max_q)
        return out

    def vit_backward(self, tape: dict, d_out: torch.Tensor, G: FlatParams, on_ready=None) -> None:
        """d_out bf16 [Nv, hidden] = gradient of the merged video embeds; accumulates into G (fp32).  ``on_ready(prefix)``
        is called when the gradients of a parameter group are final (data-parallel overlap, grpo.GradReducer)."""
        ready = on_ready or (lambda prefix: None)
        if self.cfg.vit_kind == "qwen2_5":
            return self._vit25_backward(tape, d_out, G, ready)
        cfg, W = self.cfg, self.W
        D, Hh, hd = cfg.vit_dim, cfg.vit_heads, cfg.vit_head_dim
        Np = tape["pix"].shape[0]
        scale = hd ** -0.5
        cos, sin, segs, max_q = tape["cos"], tape["sin"], tape["segs"], tape["max_q"]
        # merger
        d_g = self._dx(d_out, "merger.m2_w")
        self._dw(G["merger.m2_w"], d_out, tape["g"]); K.bias_grad_(d_out, G["merger.m2_b"])
        d_m1 = K.act_bwd(tape["m1"], d_g, K.SPACER_ACT_GELU_ERF)
        d_hm4 = self._dx(d_m1, "merger.m0_w")
        self._dw(G["merger.m0_w"], d_m1, tape["hm4"]); K.bias_grad_(d_m1, G["merger.m0_b"])
        dx = self._empty(Np, D)
        K.layernorm_bwd(tape["x_last"], W["merger.ln_w"], d_hm4.view(Np, D), tape["mean"], tape["rstd"], dx,
                        G["merger.ln_w"], G["merger.ln_b"], accumulate=False)
        ready("merger.")
        for i in reversed(range(cfg.vit_depth)):
            p = f"vit.{i}."
            t = tape["blocks"][i]
            if t["a"] is None:                                           # recompute policy: the forward's own launches
                if t.get("h2_lo") is not None:                           # precise tape: the pair launch, bit-identical to the stored policy
                    t["f1"] = self._empty(Np, W[p + "fc1_w"].shape[0], dtype=BF16)
                    t["a"] = K.gemm_pair_act(t["h2"], t["h2_lo"], W[p + "fc1_w"], K.SPACER_ACT_QUICK_GELU, bias=W[p + "fc1_b"], pre_out=t["f1"])[0]
                    t["h2_lo"] = None
                else:
                    t["f1"] = K.gemm_nt(t["h2"], W[p + "fc1_w"], bias=W[p + "fc1_b"])
                    t["a"] = K.act_fwd(t["f1"], K.SPACER_ACT_QUICK_GELU)
            dyb = K.cast_bf16(dx)
            d_a = self._dx(dyb, p + "fc2_w")
            self._dw(G[p + "fc2_w"], dyb, t["a"]); K.bias_grad_(dyb, G[p + "fc2_b"])
            d_f1 = K.act_bwd(t["f1"], d_a, K.SPACER_ACT_QUICK_GELU)
            d_h2 = self._dx(d_f1, p + "fc1_w")
            self._dw(G[p + "fc1_w"], d_f1, t["h2"]); K.bias_grad_(d_f1, G[p + "fc1_b"])
            K.layernorm_bwd(t["x_mid"], W[p + "n2_w"], d_h2, t["mean2"], t["rstd2"], dx, G[p + "n2_w"], G[p + "n2_b"])
            dyb = K.cast_bf16(dx)
            d_o = self._dx(dyb, p + "proj_w")
            self._dw(G[p + "proj_w"], dyb, t["o"]); K.bias_grad_(dyb, G[p + "proj_b"])
            qkv = t["qkv"]
            d_qkv = torch.empty_like(qkv)
            dk32, dv32 = self._zeros(Np, D), self._zeros(Np, D)
            K.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], t["o"], d_o, t["lse"], segs, max_q, Hh, Hh, hd, False,
                       scale, dq=d_qkv[:, :D], dk32=dk32, dv32=dv32)
            K.cast_bf16_strided(dk32, d_qkv[:, D:2 * D]); K.cast_bf16_strided(dv32, d_qkv[:, 2 * D:])
            K.rope_(d_qkv, cos, sin, 2 * Hh, hd, inverse=True)
            d_h = self._dx(d_qkv, p + "qkv_w")
            self._dw(G[p + "qkv_w"], d_qkv, t["h"]); K.bias_grad_(d_qkv, G[p + "qkv_b"])
            K.layernorm_bwd(t["x_in"], W[p + "n1_w"], d_h, t["mean1"], t["rstd1"], dx, G[p + "n1_w"], G[p + "n1_b"])
            tape["blocks"][i] = None
            ready(p)
        self._dw(G["vit.patch_w"], K.cast_bf16(dx), tape["pix"])
        ready("vit.patch_w")

    # ------------------------------------------------------------------ Qwen2.5-VL vision tower
    def _vit25_forward(self, pix: torch.Tensor, grids, tape: Optional[dict] = None):
        """HF Qwen2_5_VisionTransformerPretrainedModel.forward (modeling_qwen2_5_vl.py:408-472).  The window regrouping
        is applied ONCE to the input pixel rows (patch embedding is row-wise), every block then runs in the permuted
        order with windows (or whole frames on cfg.vit_fullatt blocks) as attention segments, and the merged rows are
        gathered back at the end: two bf16 row gathers instead of permuting fp32 activations."""
        cfg, W = self.cfg, self.W
        D, Hh, hd, Ip = cfg.vit_dim, cfg.vit_heads, cfg.vit_head_dim, cfg.vit_mlp_pad
        Np = pix.shape[0]
        unit_perm, row_perm, win_list = POS.vit_window_plan(grids, cfg)
        frame_list = POS.vit_segments(grids)
        rows_dev = row_perm.to(self.dev)
        cos, sin = POS.vit_tables(grids, cfg, self.dev)
        cos, sin = cos.index_select(0, rows_dev).contiguous(), sin.index_select(0, rows_dev).contiguous()
        win_segs, frame_segs = K.make_segments(win_list, self.dev), K.make_segments(frame_list, self.dev)
        max_win, max_frame = max(s[1] for s in win_list), max(s[1] for s in frame_list)
        scale = hd ** -0.5
        pix_p = K.gather_rows(pix, rows_dev.int())
        x = K.gemm_nt(pix_p, W["vit.patch_w"], out_dtype=F32)
        blocks: List[dict] = []
        for i in range(cfg.vit_depth):
            p = f"vit.{i}."
            segs, max_q = (frame_segs, max_frame) if i in cfg.vit_fullatt else (win_segs, max_win)
            rstd1 = self._empty(Np)
            h = K.rmsnorm_fwd(x, W[p + "n1_w"], 1e-6, rstd=rstd1)
            qkv = K.gemm_nt(h, W[p + "qkv_w"], bias=W[p + "qkv_b"])
            K.rope_(qkv, cos, sin, 2 * Hh, hd)
            o, lse = K.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], segs, max_q, Hh, Hh, hd, False, scale)
            x_mid = K.gemm_nt(o, W[p + "proj_w"], bias=W[p + "proj_b"], residual=x, out_dtype=F32)
            rstd2 = self._empty(Np)
            h2 = K.rmsnorm_fwd(x_mid, W[p + "n2_w"], 1e-6, rstd=rstd2)
            # [gate (Ip) | up (Ip)], padding columns are 0; SwiGLU in the GEMM epilogue, gate|up kept only for a backward pass
            keep = tape is not None and not self.recompute
            a, gu = K.gemm_swiglu(h2, W[p + "gu_w"], bias=W[p + "gu_b"], keep_gu=keep)
            x_out = K.gemm_nt(a, W[p + "down_w"], bias=W[p + "down_b"], residual=x_mid, out_dtype=F32)
            if tape is not None:
                blocks.append(dict(x_in=x, rstd1=rstd1, h=h, qkv=qkv, o=o, lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2, gu=gu,
                                   a=a if keep else None, segs=segs, max_q=max_q))
            x = x_out
        rstd = self._empty(Np)
        hm = K.rmsnorm_fwd(x, W["merger.ln_w"], 1e-6, rstd=rstd)
        m4 = cfg.merge ** 2
        hm4 = hm.view(Np // m4, m4 * D)
        m1 = K.gemm_nt(hm4, W["merger.m0_w"], bias=W["merger.m0_b"])
        g = K.act_fwd(m1, K.SPACER_ACT_GELU_ERF)
        merged = K.gemm_nt(g, W["merger.m2_w"], bias=W["merger.m2_b"])
        unit_dev = unit_perm.to(self.dev)
        out = K.gather_rows(merged, torch.argsort(unit_dev).int())           # back to the original merge-unit order
        if tape is not None:
            tape.update(pix=pix_p, blocks=blocks, x_last=x, rstd=rstd, hm4=hm4, m1=m1, g=g, cos=cos, sin=sin,
                        unit_perm=unit_dev.int())
        return out

    def _vit25_backward(self, tape: dict, d_out: torch.Tensor, G: FlatParams, ready) -> None:
        cfg, W = self.cfg, self.W
        D, Hh, hd = cfg.vit_dim, cfg.vit_heads, cfg.vit_head_dim
        Np = tape["pix"].shape[0]
        scale = hd ** -0.5
        cos, sin = tape["cos"], tape["sin"]
        # out[i] = merged[rev[i]]  =>  d_merged[u] = d_out[perm[u]]; a precise-mode tape hands d_out over in the tower's order already
        d_merged = K.gather_rows(d_out, tape["unit_perm"]) if tape["unit_perm"] is not None else d_out
        d_g = self._dx(d_merged, "merger.m2_w")
        self._dw(G["merger.m2_w"], d_merged, tape["g"]); K.bias_grad_(d_merged, G["merger.m2_b"])
        d_m1 = K.act_bwd(tape["m1"], d_g, K.SPACER_ACT_GELU_ERF)
        d_hm4 = self._dx(d_m1, "merger.m0_w")
        self._dw(G["merger.m0_w"], d_m1, tape["hm4"]); K.bias_grad_(d_m1, G["merger.m0_b"])
        dx = self._empty(Np, D)
        K.rmsnorm_bwd(tape["x_last"], W["merger.ln_w"], d_hm4.view(Np, D), tape["rstd"], dx, G["merger.ln_w"], accumulate=False)
        ready("merger.")
        for i in reversed(range(cfg.vit_depth)):
            p = f"vit.{i}."
            t = tape["blocks"][i]
            if t["a"] is None:                                           # recompute policy
                if t.get("h2_lo") is not None:                           # precise tape: the pair launch, bit-identical to the stored policy
                    t["gu"] = self._empty(Np, W[p + "gu_w"].shape[0], dtype=BF16)
                    t["a"] = K.gemm_pair_swiglu(t["h2"], t["h2_lo"], W[p + "gu_w"], bias=W[p + "gu_b"], gu_out=t["gu"])[0]
                    t["h2_lo"] = None
                else:
                    t["a"], t["gu"] = K.gemm_swiglu(t["h2"], W[p + "gu_w"], bias=W[p + "gu_b"], keep_gu=True)
            dyb = K.cast_bf16(dx)
            d_a = self._dx(dyb, p + "down_w")
            self._dw(G[p + "down_w"], dyb, t["a"]); K.bias_grad_(dyb, G[p + "down_b"])
            d_gu = K.swiglu_bwd(t["gu"], d_a)
            d_h2 = self._dx(d_gu, p + "gu_w")
            self._dw(G[p + "gu_w"], d_gu, t["h2"]); K.bias_grad_(d_gu, G[p + "gu_b"])
            K.rmsnorm_bwd(t["x_mid"], W[p + "n2_w"], d_h2, t["rstd2"], dx, G[p + "n2_w"])
            dyb = K.cast_bf16(dx)
            d_o = self._dx(dyb, p + "proj_w")
            self._dw(G[p + "proj_w"], dyb, t["o"]); K.bias_grad_(dyb, G[p + "proj_b"])
            qkv = t["qkv"]
            d_qkv = torch.empty_like(qkv)
            dk32, dv32 = self._zeros(Np, D), self._zeros(Np, D)
            K.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], t["o"], d_o, t["lse"], t["segs"], t["max_q"], Hh, Hh, hd, False,
                       scale, dq=d_qkv[:, :D], dk32=dk32, dv32=dv32)
            K.cast_bf16_strided(dk32, d_qkv[:, D:2 * D]); K.cast_bf16_strided(dv32, d_qkv[:, 2 * D:])
            K.rope_(d_qkv, cos, sin, 2 * Hh, hd, inverse=True)
            d_h = self._dx(d_qkv, p + "qkv_w")
            self._dw(G[p + "qkv_w"], d_qkv, t["h"]); K.bias_grad_(d_qkv, G[p + "qkv_b"])
            K.rmsnorm_bwd(t["x_in"], W[p + "n1_w"], d_h, t["rstd1"], dx, G[p + "n1_w"])
            tape["blocks"][i] = None
            ready(p)
        self._dw(G["vit.patch_w"], K.cast_bf16(dx), tape["pix"])
        ready("vit.patch_w")

    # ================================================================== language model
    def llm_forward(self, x: torch.Tensor, cos, sin, segs, max_q: int, *, tape: Optional[list] = None, kv_sink=None):
        """x fp32 [T, hidden] input embeddings -> fp32 [T, hidden] before the final norm."""
        cfg, W = self.cfg, self.W
        Hq, Hkv, D, I = cfg.heads, cfg.kv_heads, cfg.head_dim, cfg.intermediate
        T = x.shape[0]
        qd, kd = Hq * D, Hkv * D
        scale = D ** -0.5
        for i in range(cfg.layers):
            p = f"llm.{i}."
            rstd1 = self._empty(T)
            h = K.rmsnorm_fwd(x, W[p + "ln1_w"], cfg.rms_eps, rstd=rstd1)
            qkv = K.gemm_nt(h, W[p + "qkv_w"], bias=W[p + "qkv_b"])
            K.rope_(qkv, cos, sin, Hq + Hkv, D)
            q, k, v = qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:]
            o, lse = K.attn_fwd(q, k, v, segs, max_q, Hq, Hkv, D, True, scale)
            if kv_sink is not None:
                kv_sink(i, k, v)
            x_mid = K.gemm_nt(o, W[p + "o_w"], residual=x, out_dtype=F32)
            rstd2 = self._empty(T)
            h2 = K.rmsnorm_fwd(x_mid, W[p + "ln2_w"], cfg.rms_eps, rstd=rstd2)
            keep = tape is not None and not self.recompute
            a, gu = K.gemm_swiglu(h2, W[p + "gu_w"], keep_gu=keep)                   # SwiGLU in the gate|up GEMM's epilogue
            x_out = K.gemm_nt(a, W[p + "down_w"], residual=x_mid, out_dtype=F32)
            if tape is not None:
                tape.append(dict(x_in=x, rstd1=rstd1, h=h, qkv=qkv, o=o, lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2, gu=gu,
                                 a=a if keep else None))
            x = x_out
        return x

    def llm_backward(self, tape: list, dx: torch.Tensor, G: FlatParams, cos, sin, segs, max_q: int, on_ready=None) -> torch.Tensor:
        """dx fp32 [T, hidden] = grad of the pre-final-norm stream (updated in place); returns d(embeddings)."""
        cfg, W = self.cfg, self.W
        ready = on_ready or (lambda prefix: None)
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        T = dx.shape[0]
        qd, kd = Hq * D, Hkv * D
        scale = D ** -0.5
        for i in reversed(range(cfg.layers)):
            p = f"llm.{i}."
            t = tape[i]
            if t["a"] is None:              # recompute policy: gate|up + SwiGLU again from the saved h2, same launch, same bits
                if t.get("h2_lo") is not None:      # ... of a PRECISE forward: the pair launch on (hi, lo), whose hi half / bf16 tape it kept
                    t["gu"] = self._empty(T, W[p + "gu_w"].shape[0], dtype=BF16)
                    t["a"] = K.gemm_pair_swiglu(t["h2"], t["h2_lo"], W[p + "gu_w"], gu_out=t["gu"])[0]
                    t["h2_lo"] = None
                else:
                    t["a"], t["gu"] = K.gemm_swiglu(t["h2"], W[p + "gu_w"], keep_gu=True)
            dyb = K.cast_bf16(dx)
            d_a = self._dx(dyb, p + "down_w")
            self._dw(G[p + "down_w"], dyb, t["a"])
            d_gu = K.swiglu_bwd(t["gu"], d_a)
            d_h2 = self._dx(d_gu, p + "gu_w")
            self._dw(G[p + "gu_w"], d_gu, t["h2"])
            K.rmsnorm_bwd(t["x_mid"], W[p + "ln2_w"], d_h2, t["rstd2"], dx, G[p + "ln2_w"])
            dyb = K.cast_bf16(dx)
            d_o = self._dx(dyb, p + "o_w")
            self._dw(G[p + "o_w"], dyb, t["o"])
            qkv = t["qkv"]
            d_qkv = torch.empty_like(qkv)
            dk32, dv32 = self._zeros(T, kd), self._zeros(T, kd)
            K.attn_bwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], t["o"], d_o, t["lse"], segs, max_q, Hq, Hkv, D,
                       True, scale, dq=d_qkv[:, :qd], dk32=dk32, dv32=dv32)
            K.cast_bf16_strided(dk32, d_qkv[:, qd:qd + kd]); K.cast_bf16_strided(dv32, d_qkv[:, qd + kd:])
            K.rope_(d_qkv, cos, sin, Hq + Hkv, D, inverse=True)
            d_h = self._dx(d_qkv, p + "qkv_w")
            self._dw(G[p + "qkv_w"], d_qkv, t["h"]); K.bias_grad_(d_qkv, G[p + "qkv_b"])
            K.rmsnorm_bwd(t["x_in"], W[p + "ln1_w"], d_h, t["rstd1"], dx, G[p + "ln1_w"])
            tape[i] = None
            ready(p)
        return dx

    # ================================================================== precise mode (forward; optionally emits the backward's tape)
    # csrc/precise.hip: every activation between operators is fp32 or a (hi, lo) bf16 pair, every linear layer is two
    # accumulate passes of the production GEMM, attention runs on pair operands.  Same operator order as the fast path.
    # With ``tape``: the forward also leaves exactly what the FAST backward reads -- the fp32 residual-stream copies, the norm
    # statistics, the attention log-sum-exp, bf16 pre-activations, and the hi halves of the pairs as the bf16 activations -- so
    # ``backward_group`` runs unchanged on a tape whose log-probs hold the north-star's 1e-3 (GRPOHyper.precise_logps).
    def _vit_forward_precise(self, pix: torch.Tensor, grids, tape: Optional[dict] = None):
        """pix bf16 [Np, patch_kpad] -> (merged vision embeds fp32 [Np/4, hidden] in merge-unit order of the tower, unit_rev)."""
        cfg, W = self.cfg, self.W
        D, Hh, hd = cfg.vit_dim, cfg.vit_heads, cfg.vit_head_dim
        Np = pix.shape[0]
        v25 = cfg.vit_kind == "qwen2_5"
        taped = tape is not None
        keep = taped and not self.recompute
        cos, sin = POS.vit_tables(grids, cfg, self.dev)
        frame_list = POS.vit_segments(grids)
        frame_segs, max_frame = K.make_segments(frame_list, self.dev), max(s[1] for s in frame_list)
        unit_perm = None
        if v25:
            unit_perm, row_perm, win_list = POS.vit_window_plan(grids, cfg)
            rows_dev = row_perm.to(self.dev)
            cos, sin = cos.index_select(0, rows_dev).contiguous(), sin.index_select(0, rows_dev).contiguous()
            win_segs, max_win = K.make_segments(win_list, self.dev), max(s[1] for s in win_list)
            pix = K.gather_rows(pix, rows_dev.int())
        scale = hd ** -0.5
        stat = (lambda: self._empty(Np)) if taped else (lambda: None)
        x = K.gemm_nt(pix, W["vit.patch_w"], out_dtype=F32)                # pixel rows are exactly bf16: one pass
        blocks: List[dict] = []
        for i in range(cfg.vit_depth):
            p = f"vit.{i}."
            segs, max_q = (frame_segs, max_frame) if (not v25 or i in cfg.vit_fullatt) else (win_segs, max_win)
            mean1, rstd1 = (None if v25 else stat()), stat()
            h = K.norm_pair(x, W[p + "n1_w"], None if v25 else W[p + "n1_b"], 1e-6, mean=mean1, rstd=rstd1)
            qh, ql = K.gemm_pair_rope(*h, W[p + "qkv_w"], cos, sin, 2 * Hh, 3 * Hh, hd, bias=W[p + "qkv_b"])   # (head_dim 80: unfused)
            lse = self._empty(Hh, Np) if taped else None
            o = K.attn_fwd_pair((qh[:, :D], ql[:, :D]), (qh[:, D:2 * D], ql[:, D:2 * D]), (qh[:, 2 * D:], ql[:, 2 * D:]),
                                segs, max_q, Hh, Hh, hd, False, scale, lse=lse)
            del ql
            # taped: every residual-stream state is its own buffer (the backward reads x_in / x_mid); else updated in place
            x_mid = K.gemm_pair(*o, W[p + "proj_w"], bias=W[p + "proj_b"], residual=x, out=None if taped else x)
            mean2, rstd2 = (None if v25 else stat()), stat()
            h2 = K.norm_pair(x_mid, W[p + "n2_w"], None if v25 else W[p + "n2_b"], 1e-6, mean=mean2, rstd=rstd2)
            if v25:
                pre = self._empty(Np, W[p + "gu_w"].shape[0], dtype=BF16) if keep else None
                a = K.gemm_pair_swiglu(*h2, W[p + "gu_w"], bias=W[p + "gu_b"], gu_out=pre)   # SwiGLU + split in the pair GEMM's epilogue
                x_out = K.gemm_pair(*a, W[p + "down_w"], bias=W[p + "down_b"], residual=x_mid, out=None if taped else x_mid)
                if taped:
                    blocks.append(dict(x_in=x, rstd1=rstd1, h=h[0], qkv=qh, o=o[0], lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2[0], gu=pre,
                                       a=a[0] if keep else None, segs=segs, max_q=max_q, h2_lo=None if keep else h2[1]))
            else:
                pre = self._empty(Np, W[p + "fc1_w"].shape[0], dtype=BF16) if keep else None
                a = K.gemm_pair_act(*h2, W[p + "fc1_w"], K.SPACER_ACT_QUICK_GELU, bias=W[p + "fc1_b"], pre_out=pre)
                x_out = K.gemm_pair(*a, W[p + "fc2_w"], bias=W[p + "fc2_b"], residual=x_mid, out=None if taped else x_mid)
                if taped:
                    blocks.append(dict(x_in=x, mean1=mean1, rstd1=rstd1, h=h[0], qkv=qh, o=o[0], lse=lse, x_mid=x_mid, mean2=mean2,
                                       rstd2=rstd2, h2=h2[0], f1=pre, a=a[0] if keep else None, h2_lo=None if keep else h2[1]))
            del h, o, h2, a
            x = x_out
        mean, rstd = (None if v25 else stat()), stat()
        hm = K.norm_pair(x, W["merger.ln_w"], None if v25 else W["merger.ln_b"], 1e-6, mean=mean, rstd=rstd)
        m4 = cfg.merge ** 2
        hm4 = hm[0].view(Np // m4, m4 * D)
        m1 = self._empty(Np // m4, W["merger.m0_w"].shape[0], dtype=BF16) if taped else None
        g = K.gemm_pair_act(hm4, hm[1].view(Np // m4, m4 * D), W["merger.m0_w"], K.SPACER_ACT_GELU_ERF, bias=W["merger.m0_b"], pre_out=m1)
        merged = K.gemm_pair(*g, W["merger.m2_w"], bias=W["merger.m2_b"])
        if taped:
            # unit_perm None: the gradient of the vision rows arrives in the tower's own (window) order -- the embedding maps the
            # placeholder tokens through unit_rev instead of gathering the rows back
            tape.update(pix=pix, blocks=blocks, x_last=x, mean=mean, rstd=rstd, hm4=hm4, m1=m1, g=g[0], cos=cos, sin=sin,
                        segs=frame_segs, max_q=max_frame, unit_perm=None)
        return merged, (None if unit_perm is None else torch.argsort(unit_perm).to(self.dev))

    def _llm_forward_precise(self, x: torch.Tensor, cos, sin, segs, max_q: int, tape: Optional[list] = None) -> torch.Tensor:
        """x fp32 [T, hidden] -> the stream before the final norm (updated in place layer by layer unless taped)."""
        cfg, W = self.cfg, self.W
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        qd, kd = Hq * D, Hkv * D
        T = x.shape[0]
        scale = D ** -0.5
        taped = tape is not None
        keep = taped and not self.recompute
        for i in range(cfg.layers):
            p = f"llm.{i}."
            rstd1 = self._empty(T) if taped else None
            h = K.norm_pair(x, W[p + "ln1_w"], None, cfg.rms_eps, rstd=rstd1)
            # bias + rotary + hi/lo split in the q|k|v pair GEMM's epilogue: no fp32 [T, qkv] round trip
            qh, ql = K.gemm_pair_rope(*h, W[p + "qkv_w"], cos, sin, Hq + Hkv, Hq + 2 * Hkv, D, bias=W[p + "qkv_b"])
            lse = self._empty(Hq, T) if taped else None
            o = K.attn_fwd_pair((qh[:, :qd], ql[:, :qd]), (qh[:, qd:qd + kd], ql[:, qd:qd + kd]), (qh[:, qd + kd:], ql[:, qd + kd:]),
                                segs, max_q, Hq, Hkv, D, True, scale, lse=lse)
            del ql
            x_mid = K.gemm_pair(*o, W[p + "o_w"], residual=x, out=None if taped else x)
            rstd2 = self._empty(T) if taped else None
            h2 = K.norm_pair(x_mid, W[p + "ln2_w"], None, cfg.rms_eps, rstd=rstd2)
            gu = self._empty(T, W[p + "gu_w"].shape[0], dtype=BF16) if keep else None
            a = K.gemm_pair_swiglu(*h2, W[p + "gu_w"], gu_out=gu)       # SwiGLU + split (+ bf16 gate|up tape) in the pair GEMM's epilogue
            x_out = K.gemm_pair(*a, W[p + "down_w"], residual=x_mid, out=None if taped else x_mid)
            if taped:
                # recompute policy: the backward re-runs THIS layer's gate|up launch, which needs the lo half of its input as well
                tape.append(dict(x_in=x, rstd1=rstd1, h=h[0], qkv=qh, o=o[0], lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2[0], gu=gu,
                                 a=a[0] if keep else None, h2_lo=None if keep else h2[1]))
            del h, o, h2, a
            x = x_out
        return x

    # ================================================================== embeddings
    def embed(self, ids: torch.Tensor, video: Optional[torch.Tensor], placeholder_scopes: Optional[Sequence[Tuple[int, int]]] = None,
              unit_rev: Optional[torch.Tensor] = None):
        """ids int64 [T] (device); video bf16 [Nv, hidden] rows replace placeholder tokens in order.  Only tokens inside
        ``placeholder_scopes`` ([start, end) ranges: the prompts) can be placeholders: a SAMPLED completion token that happens
        to be <|video_pad|> / <|image_pad|> (random-init policies do emit them) is an ordinary token with its own embedding row
        -- HF raises on the count mismatch there and the reference falls back to a text-only forward (TR:526-532)."""
        cfg = self.cfg
        vrow = None
        if video is not None:
            is_vis = (ids == cfg.video_token_id) | (ids == cfg.image_token_id)
            if placeholder_scopes is not None:
                inside = torch.zeros_like(is_vis)
                for a, b in placeholder_scopes:
                    inside[a:b] = True
                is_vis &= inside
            vrow = torch.where(is_vis, torch.cumsum(is_vis.int(), 0, dtype=torch.int32) - 1,
                               torch.full_like(ids, -1, dtype=torch.int32)).int().contiguous()
            if unit_rev is not None:       # vision rows still in the window order of the Qwen2.5 tower: row i lives at unit_rev[i]
                vrow = torch.where(vrow >= 0, unit_rev.int()[vrow.clamp(min=0).long()], vrow).int().contiguous()
        if video is not None and video.dtype == F32:                        # precise scoring mode
            return K.embed_fwd_f32video(ids, self.W["llm.embed"], video, vrow), vrow
        return K.embed_fwd(ids, self.W["llm.embed"], video, vrow), vrow

    # ================================================================== group scoring (policy / reference logps)
    @staticmethod
    def group_layout(P: int, Kn: int, C: int):
        """Token layout [prompt | comp_0 | ... | comp_{K-1}] -> (segments, rows whose logits predict completions)."""
        segs = [(0, P, 0, 0)] + [(P + k * C, C, 0, P) for k in range(Kn)]
        t = torch.arange(C)
        sel = torch.stack([torch.where(t == 0, torch.full_like(t, P - 1), P + k * C + t - 1) for k in range(Kn)])
        return segs, sel.reshape(-1).int()

    def score_group(self, prompt_ids: torch.Tensor, completion_ids: torch.Tensor, pix: Optional[torch.Tensor], grids,
                    *, tape: Optional[dict] = None, era_rule: bool = False, precise: bool = False,
                    lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """Per-token log-probs [K, C] of K completions of one prompt (SG_RLVR_trainer.py:353-366,527-528),
        computed with the prompt shared: the prompt runs once and every rollout attends its keys.  ``precise=True``: the
        forward-only mode that holds the north-star's 1e-3 against an fp32 evaluation at full depth (csrc/precise.hip)."""
        return self.score_groups([(prompt_ids, pix, grids)], [completion_ids], tape=tape, era_rule=era_rule, precise=precise, lengths=lengths)

    def score_groups(self, prompts: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], Optional[Sequence]]],
                     completions: Sequence[torch.Tensor], *, tape: Optional[dict] = None, era_rule: bool = False,
                     precise: bool = False, prefill: Optional[Sequence] = None, lengths: Optional[Sequence[int]] = None) -> torch.Tensor:
        """``score_group`` for SEVERAL prompt groups in ONE token-packed pass: prompts[g] = (prompt_ids, pix, grids),
        completions[g] int64 [K, C] (same K, C for all g); returns [G*K, C] in group order.  Groups are independent (their
        segments never see each other), so the numbers are the single-group ones; what changes is the launch shape: every GEMM
        sees G x the rows (fewer, fuller launches: the dW GEMMs' fp32 read-modify-write of the gradient is paid once per pass,
        the ViT's small-K GEMMs fill the chip) at G x the activation memory (27 GB per 7B group).
        Row layout of the pass (round 5): [prompt_0 | .. | prompt_{G-1} | completions of group 0 | .. | of group G-1] -- all prompt
        rows first, so the completion rows are ONE contiguous block.  ``prefill`` (one ``PrefillSlice`` per group, from
        ``RolloutEngine.generate`` with ``keep_prefill_tape``): the rollout's prefill already ran the ViT and the prompt rows of THIS
        policy with a tape -- the pass then computes the completion rows only and takes the prompt rows of every taped tensor from
        that tape (SG_RLVR_trainer.py:463 then :517-541 run the same prompt-side forward twice; here it runs once).
        ``lengths`` (round 6; host ints, one per rollout in group order = ``spacer_completion_mask``'s lengths: tokens up to and
        including the first EOS): EOS-TRIMMED scoring.  The reference multiplies its completion mask into the loss (TR:493-498,
        :640-643), so positions behind a rollout's first EOS contribute exactly zero to loss, KL and every gradient; the pass then
        packs only the first lengths[i] tokens of rollout i (segment i has q_len = lengths[i]), scores only those positions and
        returns 0 for the others.  Row-wise kernels, GEMM rows (K-split tail off) and the attention rows do not depend on the other
        rows of a launch, so the scored log-probs are the rectangular pass's bit for bit (tests/test_ragged_gpu.py)."""
        cfg = self.cfg
        Kn, C = completions[0].shape
        G = len(prompts)
        assert all(tuple(c.shape) == (Kn, C) for c in completions) and G == len(completions)
        if lengths is not None:
            lens = [min(C, max(1, int(n))) for n in lengths]
            assert len(lens) == G * Kn, "one length per rollout"
            if all(n == C for n in lens):
                lens = None                               # nothing to trim: the rectangular layout as it is
        else:
            lens = None
        with_video = [g for g, (_, pix, _) in enumerate(prompts) if pix is not None]
        assert len(with_video) in (0, G), "groups of one pass either all carry vision inputs or none does"
        reuse = (prefill is not None and tape is not None and not precise and all(pf is not None for pf in prefill)
                 and self._prefill_usable(prefill, prompts, era_rule))
        vit_tape = {} if tape is not None else None
        video, all_grids, unit_rev = None, [], None
        if with_video:
            all_grids = [g for _, _, gr in prompts for g in gr]
            if reuse:
                video = None                              # the prompt rows' embeddings come from the prefill tape
                vit_tape = self._vit_tape_slice(prefill)
            else:
                pix_all = prompts[0][1] if G == 1 else torch.cat([p[1] for p in prompts], 0)
                if precise:
                    video, unit_rev = self._vit_forward_precise(pix_all, all_grids, vit_tape)
                else:
                    video = self.vit_forward(pix_all, all_grids, vit_tape)
        plen = [p[0].numel() for p in prompts]
        poff = [0]
        for P in plen[:-1]:
            poff.append(poff[-1] + P)
        Pt = sum(plen)                                                     # prompt rows of the pass; completions start here
        pos_prompt, pos_comp, seg_list, sel_parts, scope, pack_parts = [], [], [], [], [], []
        t = torch.arange(C)
        c0 = Pt                                                            # next completion row of the pass
        for g, ((prompt_ids, _, grids), comp) in enumerate(zip(prompts, completions)):
            P = plen[g]
            pos3, delta = POS.mrope_positions(prompt_ids.tolist(), list(grids or []), cfg, era_rule)
            comp_pos = (P + delta) + t                                     # same for every rollout of the group
            pos_prompt.append(pos3)
            seg_list.append((poff[g], P, 0, 0))
            scope.append((poff[g], poff[g] + P))
            for k in range(Kn):
                n = C if lens is None else lens[g * Kn + k]               # tokens of this rollout in the pass (all C, or up to its EOS)
                pos_comp.append(comp_pos[:n].view(1, n).expand(3, n))
                seg_list.append((c0, n, poff[g], P))
                # rows whose logits predict the completion: the prompt's last row for token 0, then the completion rows
                sel_parts.append(torch.cat([torch.tensor([poff[g] + P - 1]), c0 + t[:n - 1]]))
                if lens is not None:
                    pack_parts.append((g * Kn + k) * C + t[:n])
                c0 += n
        T = c0
        comp_flat = torch.cat([c.reshape(-1) for c in completions]).contiguous()
        if lens is None:
            pack_idx, targets = None, comp_flat
        else:
            pack_idx = torch.cat(pack_parts).to(self.dev)                  # flat [G*K, C] positions of the packed tokens, int64
            targets = comp_flat.index_select(0, pack_idx)
        ids = torch.cat([p[0].reshape(-1) for p in prompts] + [targets])
        cos, sin = POS.mrope_tables(torch.cat(pos_prompt + pos_comp, dim=1), cfg, self.dev)
        sel = torch.cat(sel_parts).int().to(self.dev)
        max_q = max(s[1] for s in seg_list)
        max_c = max(s[1] for s in seg_list if s[3] > 0)
        llm_tape = [] if tape is not None else None
        if reuse:
            # placeholders sit in the prompt rows only: the vision-row map for embed_bwd is computed from the ids, no embedding of them
            _, vrow = self._vision_rows(ids, scope) if with_video else (None, None)
            xc = K.embed_fwd(ids[Pt:], self.W["llm.embed"], None, None)      # completion rows are ordinary tokens
            comp_segs = [s for s in seg_list if s[3] > 0]
            x = self._llm_forward_reuse(xc, cos, sin, K.make_segments(comp_segs, self.dev), max_c, Pt, prefill, llm_tape)
            segs = K.make_segments(seg_list, self.dev)
        else:
            segs = K.make_segments(seg_list, self.dev)
            if precise:
                x0, vrow = self.embed(ids, video, placeholder_scopes=scope, unit_rev=unit_rev)
                x = self._llm_forward_precise(x0, cos, sin, segs, max_q, tape=llm_tape)
            else:
                x0, vrow = self.embed(ids, video, placeholder_scopes=scope)
                x = self.llm_forward(x0, cos, sin, segs, max_q, tape=llm_tape)
        logp = self.head_forward(x, sel, targets, tape, precise=precise)
        if tape is not None:
            tape.update(vit=vit_tape, llm=llm_tape, ids=ids, vrow=vrow, cos=cos, sin=sin, segs=segs, max_q=max_q, T=T,
                        has_video=bool(with_video), reused_prefill=reuse, pack_idx=pack_idx)
        if pack_idx is None:
            return logp.view(G * Kn, C)
        return K.scatter_f32_(logp, pack_idx, self._zeros(G * Kn, C))      # 0 behind the mask (the loss kernel multiplies by it)

    # ------------------------------------------------------------------ prompt-side forward taken from the rollout's prefill
    def _vision_rows(self, ids: torch.Tensor, scopes):
        """(None, vrow): the token -> vision-row map of ``embed`` without embedding anything."""
        cfg = self.cfg
        is_vis = (ids == cfg.video_token_id) | (ids == cfg.image_token_id)
        inside = torch.zeros_like(is_vis)
        for a, b in scopes:
            inside[a:b] = True
        is_vis &= inside
        vrow = torch.where(is_vis, torch.cumsum(is_vis.int(), 0, dtype=torch.int32) - 1, torch.full_like(ids, -1, dtype=torch.int32)).int().contiguous()
        return None, vrow

    def _prefill_usable(self, prefill, prompts, era_rule) -> bool:
        """The slices must come from ONE prefill pass of this engine's current weights, cover consecutive prompts of that pass in
        order, be taped with the stored policy, and describe the same prompts."""
        first = prefill[0]
        sh = first.shared
        if self.recompute or self.cfg.vit_kind != "qwen2" or (sh.get("engine") or (lambda: None))() is not self or sh.get("era_rule") != era_rule:
            return False
        if sh.get("weights_version") != self.weights_version:
            return False
        for j, (pf, pr) in enumerate(zip(prefill, prompts)):
            if pf.shared is not sh or pf.index != first.index + j or pf.P != pr[0].numel():
                return False
            if (pr[1] is None) != (pf.n_patch == 0):
                return False
        return True

    @staticmethod
    def _rows(arr: Optional[torch.Tensor], a: int, b: int):
        return None if arr is None else arr[a:b]

    def _vit_tape_slice(self, prefill) -> dict:
        """The vision tower's tape of the pass = row slices (views) of the prefill's tape: patches [a, b) of the packed pass."""
        sh = prefill[0].shared
        vt = sh["vit"]
        a, b = prefill[0].patch0, prefill[-1].patch0 + prefill[-1].n_patch
        m4 = self.cfg.merge ** 2
        R = self._rows
        blocks = []
        for t in vt["blocks"]:
            # (lse is [heads, patches]: attn_bwd reads it with a row stride of the pass's patch count, so its column slice is copied)
            blocks.append({k: (v[:, a:b].contiguous() if k == "lse" else R(v, a, b)) for k, v in t.items()})
        seg_list = [(qs - a, ql, 0, 0) for (qs, ql, _, _) in sh["vit_segments"] if a <= qs < b]
        out = dict(pix=R(vt["pix"], a, b), blocks=blocks, x_last=R(vt["x_last"], a, b), mean=R(vt["mean"], a, b), rstd=R(vt["rstd"], a, b),
                   hm4=R(vt["hm4"], a // m4, b // m4), m1=R(vt["m1"], a // m4, b // m4), g=R(vt["g"], a // m4, b // m4),
                   cos=R(vt["cos"], a, b), sin=R(vt["sin"], a, b), segs=K.make_segments(seg_list, self.dev), max_q=vt["max_q"])
        return out

    def _llm_forward_reuse(self, xc: torch.Tensor, cos, sin, comp_segs, max_q: int, Pt: int, prefill, tape: list) -> torch.Tensor:
        """``llm_forward`` (taped) of a pass whose prompt rows [0, Pt) were already computed and taped by the rollout's prefill: only
        the completion rows [Pt, T) run through the layer's kernels; they attend the prompts' keys / values, which are copied from
        the prefill tape together with the prompt rows of every other taped tensor (the backward reads whole [T, features] arrays;
        GEMM outputs land directly in the completion rows of those arrays).  Row-wise kernels and GEMM rows do not depend on the
        other rows of a launch, so every number equals the full pass's bit for bit when the GEMM's K-split tail is off.  Returns
        the pre-final-norm stream [T, hidden]."""
        cfg, W = self.cfg, self.W
        Hq, Hkv, D = cfg.heads, cfg.kv_heads, cfg.head_dim
        qd, kd = Hq * D, Hkv * D
        Tc = xc.shape[0]
        T = Pt + Tc
        scale = D ** -0.5
        sh = prefill[0].shared
        r0, r1 = prefill[0].row0, prefill[-1].row0 + prefill[-1].P
        assert r1 - r0 == Pt
        cos_c, sin_c = cos[Pt:], sin[Pt:]

        def full(cached: torch.Tensor):
            """[T, F] buffer whose prompt rows are the prefill's; returns (buffer, the view of its completion rows)."""
            buf = torch.empty(T, *cached.shape[1:], device=self.dev, dtype=cached.dtype)
            buf[:Pt].copy_(cached[r0:r1])
            return buf, buf[Pt:]

        x_in, x = full(sh["llm"][0]["x_in"])
        x.copy_(xc)
        for i in range(cfg.layers):
            p = f"llm.{i}."
            c = sh["llm"][i]
            rstd1, rstd1_c = full(c["rstd1"])
            h, h_c = full(c["h"])
            K.rmsnorm_fwd(x, W[p + "ln1_w"], cfg.rms_eps, rstd=rstd1_c, out=h_c)
            qkv, qkv_c = full(c["qkv"])                                     # the prompts' post-rotary keys / values
            K.gemm_nt(h_c, W[p + "qkv_w"], bias=W[p + "qkv_b"], out=qkv_c)
            K.rope_(qkv_c, cos_c, sin_c, Hq + Hkv, D)
            o, o_c = full(c["o"])
            lse = self._empty(Hq, T)
            lse[:, :Pt].copy_(c["lse"][:, r0:r1])
            K.attn_fwd(qkv[:, :qd], qkv[:, qd:qd + kd], qkv[:, qd + kd:], comp_segs, max_q, Hq, Hkv, D, True, scale, out=o, lse=lse)
            x_mid, x_mid_c = full(c["x_mid"])
            rstd2, rstd2_c = full(c["rstd2"])
            h2, h2_c = full(c["h2"])
            K.gemm_nt(o_c, W[p + "o_w"], residual=x, out=x_mid_c, out_dtype=F32)
            K.rmsnorm_fwd(x_mid_c, W[p + "ln2_w"], cfg.rms_eps, rstd=rstd2_c, out=h2_c)
            gu, gu_c = full(c["gu"])
            a, a_c = full(c["a"])
            K.gemm_swiglu(h2_c, W[p + "gu_w"], keep_gu=True, out=a_c, gu_out=gu_c)
            # the layer's output goes straight into the completion rows of the NEXT layer's x_in (or of the final stream)
            nxt, nxt_c = full(sh["llm"][i + 1]["x_in"] if i + 1 < cfg.layers else sh["x_final"])
            K.gemm_nt(a_c, W[p + "down_w"], residual=x_mid_c, out=nxt_c, out_dtype=F32)
            tape.append(dict(x_in=x_in, rstd1=rstd1, h=h, qkv=qkv, o=o, lse=lse, x_mid=x_mid, rstd2=rstd2, h2=h2, gu=gu, a=a))
            x_in, x = nxt, nxt_c
        return x_in

    # ================================================================== head: final norm -> lm_head -> log-prob of the targets
    def head_forward(self, x: torch.Tensor, sel: torch.Tensor, targets: torch.Tensor, tape: Optional[dict] = None, *,
                     precise: bool = False) -> torch.Tensor:
        """x fp32 [T, hidden] (pre-final-norm stream); rows ``sel`` (int32) predict ``targets`` (int64): the reference's
        ``log_softmax(logits)[token]`` (TR:353-366) on the completion rows only.  Returns logp fp32 [len(sel)].  ``precise``: the
        normed rows travel as a (hi, lo) pair and every vocabulary chunk is a two-pass GEMM (csrc/precise.hip)."""
        cfg = self.cfg
        rstd_f = self._empty(x.shape[0])
        hsel_lo = None
        if precise:
            hn = K.norm_pair(x, self.W["llm.norm_w"], None, cfg.rms_eps, rstd=rstd_f)
            hsel, hsel_lo = K.gather_rows(hn[0], sel), K.gather_rows(hn[1], sel)
        else:
            hn = K.rmsnorm_fwd(x, self.W["llm.norm_w"], cfg.rms_eps, rstd=rstd_f)
            hsel = K.gather_rows(hn, sel)
        del hn
        # lm_head over vocabulary chunks with an online log-sum-exp: a chunk's fp32 logits update the running
        # (max, sum-exp, target logit) of every row and are dead afterwards.  Only a taped pass WITHOUT the recompute policy keeps
        # them (column slices of one [rows, vocab] buffer) for the backward.
        Wlm = self.W["llm.lm_head"]
        rows, V, ch = hsel.shape[0], Wlm.shape[0], self.HEAD_CHUNK
        store = tape is not None and not self.recompute
        logits = self._empty(rows, V) if store else None
        buf = None if store else self._empty(rows, min(ch, V))
        state = self._empty(3, rows)
        for c0 in range(0, V, ch):
            c1 = min(V, c0 + ch)
            lg = logits[:, c0:c1] if store else buf[:, :c1 - c0]
            if precise:
                K.gemm_pair(hsel, hsel_lo, Wlm[c0:c1], out=lg)
            else:
                K.gemm_nt(hsel, Wlm[c0:c1], out=lg, out_dtype=F32)
            K.lse_chunk_(lg, targets, c0, state, first=c0 == 0)
        logp, lse = K.lse_finish(state)
        if tape is not None:
            tape.update(sel=sel, x_final=x, rstd_f=rstd_f, hsel=hsel, hsel_lo=hsel_lo, logits=logits, targets=targets, lse=lse)
        return logp

    def head_backward(self, tape: dict, dlogp: torch.Tensor, G: FlatParams) -> torch.Tensor:
        """d loss / d logp (fp32 [rows]) -> gradient of the pre-final-norm stream (fp32 [T, hidden]); lm_head and final-norm
        gradients accumulate into G."""
        cfg, W = self.cfg, self.W
        T, H = tape["x_final"].shape[0], cfg.hidden
        # chunk by chunk over the vocabulary: dlogits of a chunk (from the stored chunk logits, or from logits
        # recomputed by the forward's own GEMM launch under the recompute policy) feed that chunk's dX / dW GEMMs and are dead;
        # neither [rows, vocab] tensor exists.  dX accumulates over chunks in fp32.
        Wlm, Glm = W["llm.lm_head"], G["llm.lm_head"]
        hsel, targets, lse, logits = tape["hsel"], tape["targets"], tape["lse"], tape["logits"]
        rows, V, ch = hsel.shape[0], Wlm.shape[0], self.HEAD_CHUNK
        g = dlogp.reshape(-1).contiguous()
        if tape.get("pack_idx") is not None:            # EOS-trimmed pass: d loss / d logp of the packed positions (0 elsewhere anyway)
            g = K.gather_f32(g, tape["pack_idx"])
        assert g.numel() == rows, (g.numel(), rows)
        buf = None if logits is not None else self._empty(rows, min(ch, V))
        dl_buf = self._empty(rows, min(ch, V), dtype=BF16)
        d_hsel32 = self._empty(rows, H)
        for c0 in range(0, V, ch):
            c1 = min(V, c0 + ch)
            if logits is not None:
                lg = logits[:, c0:c1]
            elif tape.get("hsel_lo") is not None:       # recompute policy behind a precise forward: the same two-pass chunk GEMM
                lg = K.gemm_pair(hsel, tape["hsel_lo"], Wlm[c0:c1], out=buf[:, :c1 - c0])
            else:
                lg = K.gemm_nt(hsel, Wlm[c0:c1], out=buf[:, :c1 - c0], out_dtype=F32)
            dl = K.logprob_bwd_chunk(lg, targets, c0, lse, g, dl_buf[:, :c1 - c0])
            K.gemm(dl, Wlm[c0:c1], trans_b=True, out=d_hsel32, residual=d_hsel32 if c0 else None, out_dtype=F32)
            self._dw(Glm[c0:c1], dl, hsel)
        tape["logits"] = None
        d_hsel = K.cast_bf16(d_hsel32)
        del d_hsel32, dl_buf, buf
        d_hn32 = self._zeros(T, H)
        K.scatter_add_rows_(d_hsel, tape["sel"], d_hn32)
        d_hn = K.cast_bf16(d_hn32)
        dx = d_hn32                                                     # reuse the buffer for the stream gradient
        K.rmsnorm_bwd(tape["x_final"], W["llm.norm_w"], d_hn, tape["rstd_f"], dx, G["llm.norm_w"], accumulate=False)
        return dx

    def score_sequence(self, ids: torch.Tensor, pix: Optional[torch.Tensor], grids, *, tape: Optional[dict] = None,
                       era_rule: bool = False, second_per_grid_ts=None) -> torch.Tensor:
        """Teacher-forced log-probs [S-1] of ids[1:] for ONE full sequence (placeholders anywhere, 3-D positions from the
        whole sequence): the quantity the SFT objective averages (open_r1/sft.py: HF causal-LM loss on ``labels``).  The tape
        has the layout ``backward_group`` consumes, with dlogp of shape [1, S-1]."""
        cfg = self.cfg
        ids = ids.reshape(-1)
        S = ids.numel()
        vit_tape = {} if tape is not None else None
        video = self.vit_forward(pix, grids, vit_tape) if pix is not None else None
        x0, vrow = self.embed(ids, video)
        pos3, _ = POS.mrope_positions(ids.tolist(), list(grids or []), cfg, era_rule, second_per_grid_ts)
        cos, sin = POS.mrope_tables(pos3, cfg, self.dev)
        segs = K.make_segments([(0, S, 0, 0)], self.dev)
        llm_tape = [] if tape is not None else None
        x = self.llm_forward(x0, cos, sin, segs, S, tape=llm_tape)
        sel = torch.arange(S - 1, device=self.dev, dtype=torch.int32)
        logp = self.head_forward(x, sel, ids[1:].contiguous(), tape)
        if tape is not None:
            tape.update(vit=vit_tape, llm=llm_tape, ids=ids, vrow=vrow, cos=cos, sin=sin, segs=segs, max_q=S, T=S,
                        has_video=video is not None)
        return logp

    def backward_group(self, tape: dict, dlogp: torch.Tensor, G: FlatParams, on_ready=None) -> None:
        """Back-propagates d loss / d logp (fp32 [K, C]) through lm_head, the LLM, the embeddings and the ViT.
        ``on_ready(prefix)``: see vit_backward."""
        cfg, W = self.cfg, self.W
        ready = on_ready or (lambda prefix: None)
        tied = cfg.tie_embeddings
        H = cfg.hidden
        dx = self.head_backward(tape, dlogp, G)
        ready("llm.norm_w")
        if not tied:
            ready("llm.lm_head")
        dx = self.llm_backward(tape["llm"], dx, G, tape["cos"], tape["sin"], tape["segs"], tape["max_q"], on_ready)
        d_video = None
        if tape["has_video"]:
            nv = int((tape["vrow"] >= 0).sum())
            d_video = self._zeros(nv, H)
        K.embed_bwd(tape["ids"], tape["vrow"], dx, G["llm.embed"], d_video)
        ready("llm.embed")
        if d_video is not None:
            self.vit_backward(tape["vit"], K.cast_bf16(d_video), G, on_ready)
        else:
            # a text-only last micro-batch: the vision gradients accumulated by earlier micro-batches are final too.  Report them in
            # the order vit_backward does, so that the data-parallel reducer issues the SAME sequence of collectives on every rank
            # whatever its data (a rank with a vision row and a rank without would otherwise disagree on bucket boundaries)
            ready("merger.")
            for i in reversed(range(cfg.vit_depth)):
                ready(f"vit.{i}.")
            ready("vit.patch_w")
