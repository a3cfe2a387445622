"""GPU: LAYER-LOCAL oracle parity at the benchmark's own widths and shapes (Qwen2-VL-7B: hidden 3584, 28 / 4 heads of 128,
intermediate 18944, vocabulary 152064; ViT 1280 x 16 heads of 80; BASELINE.json configs[2] = cfg3: one prompt group = 1402 prompt
tokens (1040 video + 362 text) + K = 8 completions of C = 512 -> T = 5498 packed tokens, grid (8, 20, 26)).

Whole-model comparisons at depth (tests/test_depth_gpu.py) measure the 28-layer AMPLIFICATION of rounding noise as much as the
kernels; here each piece is compared with the fp32 oracle (oracle/qwen2vl_fp32.py, run on the host cores) WITHOUT depth
amplification, fed the SAME fp32 inputs, forward and backward:
  * one decoder layer over the real shared-prefix segment layout (prompt causal + 8 rollouts attending prompt + own keys);
  * final norm + lm_head + log-prob of the target (SG_RLVR_trainer.py:353-366) on the 4096 completion rows;
  * a one-block vision tower (patch embed, block, merger) on the 16-frame grid.
Tolerances are bf16-operand budgets of ONE operator chain: every GEMM / attention operand is rounded to bf16 (relative 2^-9 per
element, random sign), so an output that is a K-term dot product carries a relative error of ~2^-9 of its own rms; three to four
such stages in series give ~1e-2 of the layer's output change.  The precise mode (csrc/precise.hip) must be two orders below."""
import dataclasses
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_path as CP                                      # noqa: E402
from oracle import qwen2vl_fp32 as O                                   # noqa: E402
from spacer_amd import kernels as K                                    # noqa: E402
from spacer_amd.qwen2vl import positions as POS                        # noqa: E402
from spacer_amd.qwen2vl.config import QWEN2_VL_7B                      # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine                    # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, random_init_   # noqa: E402
from spacer_amd.synthetic import make_prompt                           # noqa: E402

F32 = torch.float32
KN, C = 8, 512                       # cfg3: K = 8 rollouts of 512 tokens


def rms(t):
    return float(t.double().pow(2).mean().sqrt())


def rel_fro(got, want):
    return float((got.double() - want.double()).norm() / (want.double().norm() + 1e-30))


@pytest.fixture(scope="module")
def one(dev):
    """7B widths, ONE decoder layer + ONE vision block, the benchmark's seeded random init (N(0, 0.02), seed 1234)."""
    cfg = dataclasses.replace(QWEN2_VL_7B, layers=1, vit_depth=1)
    params = FlatParams.empty(cfg, dev)
    random_init_(params, seed=1234)
    eng = Qwen2VLEngine(cfg, params)
    w = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    w["visual.patch_embed.proj.weight"] = w["visual.patch_embed.proj.weight"].reshape(cfg.vit_dim, -1)
    prompt, frames = make_prompt(cfg, 0, 16, 280, 364, 360, dev)          # cfg3 prompt: 16 frames 280 x 364 + 360 text tokens
    torch.set_num_threads(min(32, torch.get_num_threads()))
    yield dict(cfg=cfg, ocfg=cfg.as_oracle_dict(), params=params, eng=eng, w=w, prompt=prompt, frames=frames)
    del eng, params
    torch.cuda.empty_cache()


def _layout(o, dev):
    """The packed scoring layout of one cfg3 group exactly as Qwen2VLEngine.score_groups builds it, and the oracle's twin."""
    cfg, pr = o["cfg"], o["prompt"]
    P = pr.ids.numel()
    pos3, delta = POS.mrope_positions(pr.ids.tolist(), list(pr.grids), cfg, False)
    comp_pos = (P + delta) + torch.arange(C)
    pos = torch.cat([pos3] + [comp_pos.view(1, C).expand(3, C)] * KN, dim=1)
    seg_list, sel = Qwen2VLEngine.group_layout(P, KN, C)
    cos, sin = POS.mrope_tables(pos, cfg, dev)
    opos3, odelta = O.mrope_position_ids(pr.ids.tolist(), list(pr.grids), o["ocfg"])
    assert torch.equal(opos3, pos3) and odelta == delta
    return dict(P=P, T=P + KN * C, pos=pos, segs=K.make_segments(seg_list, dev), max_q=max(s[1] for s in seg_list), sel=sel.to(dev),
                cos=cos, sin=sin, mask=CP.group_mask(P, KN, C))


def test_decoder_layer_forward_backward_at_7b_width_on_the_cfg3_layout(one, dev):
    o = one
    cfg, eng, w = o["cfg"], o["eng"], o["w"]
    L = _layout(o, dev)
    T, H = L["T"], cfg.hidden
    assert (L["P"], T) == (1402, 5498)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(T, H, generator=g)                       # a residual stream of unit scale
    dy = torch.randn(T, H, generator=g) * 0.1
    names = ["model.layers.0.self_attn.q_proj.weight", "model.layers.0.self_attn.k_proj.weight", "model.layers.0.self_attn.v_proj.weight",
             "model.layers.0.self_attn.q_proj.bias", "model.layers.0.self_attn.o_proj.weight", "model.layers.0.mlp.gate_proj.weight",
             "model.layers.0.mlp.up_proj.weight", "model.layers.0.mlp.down_proj.weight", "model.layers.0.input_layernorm.weight",
             "model.layers.0.post_attention_layernorm.weight"]
    for n in names:
        w[n].requires_grad_(True)
    x0r = x0.clone().requires_grad_(True)
    t0 = time.time()
    want = O.llm_forward(w, o["ocfg"], x0r, L["pos"], L["mask"], final_norm=False)
    (want * dy).sum().backward()
    t_oracle = time.time() - t0
    want = want.detach()
    # ---- engine, fast path
    G = o["params"].like(F32)
    tape = []
    x = eng.llm_forward(x0.to(dev), L["cos"], L["sin"], L["segs"], L["max_q"], tape=tape)
    dx = eng.llm_backward(tape, dy.to(dev).clone(), G, L["cos"], L["sin"], L["segs"], L["max_q"])
    dlt = want - x0
    e = x.cpu() - want
    print(f"decoder layer, T = {T}: |out - oracle| rms {rms(e):.2e} max {float(e.abs().max()):.2e}; layer delta rms {rms(dlt):.2e} "
          f"max {float(dlt.abs().max()):.2e}; oracle fwd+bwd {t_oracle:.1f} s on the host")
    assert rms(e) <= 6e-3 * rms(dlt), (rms(e), rms(dlt))              # measured 4.1e-3: ~1 bf16 ulp (2^-8) of the layer's output change
    assert float(e.abs().max()) <= 1e-2 * float(dlt.abs().max())          # measured 4.9e-3
    got = export_state_dict(G)
    print(f"   d input: rel Frobenius err {rel_fro(dx.cpu(), x0r.grad):.3e}")
    assert rel_fro(dx.cpu(), x0r.grad) <= 8e-3                            # measured 4.6e-3
    for n in names:
        r = rel_fro(got[n].float().cpu(), w[n].grad)
        print(f"   grad {n:50s} rel Frobenius err {r:.3e}")
        assert r <= 1.4e-2, (n, r)                                        # measured 4.4e-3 (MLP) .. 8.7e-3 (q / k: two bf16 stages more)
    for n in names:
        w[n].requires_grad_(False); w[n].grad = None
    # ---- engine, precise mode on the same inputs
    xp = eng._llm_forward_precise(x0.to(dev).clone(), L["cos"], L["sin"], L["segs"], L["max_q"])
    ep = xp.cpu() - want
    print(f"   precise mode: rms {rms(ep):.2e} max {float(ep.abs().max()):.2e}")
    assert rms(ep) <= 1e-4 * rms(dlt) and float(ep.abs().max()) <= 1e-3 * float(dlt.abs().max())


def test_head_logprobs_forward_backward_on_the_4096_completion_rows(one, dev):
    o = one
    cfg, eng, w = o["cfg"], o["eng"], o["w"]
    L = _layout(o, dev)
    T, H = L["T"], cfg.hidden
    g = torch.Generator().manual_seed(12)
    x = torch.randn(T, H, generator=g) * 3.0                   # a late-layer stream: the final norm has work to do
    targets = torch.randint(0, cfg.vocab, (KN * C,), generator=g)
    dlogp = torch.randn(KN * C, generator=g) * 0.1
    sel_h = L["sel"].cpu().long()
    names = ["model.norm.weight", "lm_head.weight"]
    for n in names:
        w[n].requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    t0 = time.time()
    hn = O.rms_norm(xr, w["model.norm.weight"], cfg.rms_eps)
    lg = hn[sel_h] @ w["lm_head.weight"].t()
    want = torch.log_softmax(lg, -1).gather(1, targets.view(-1, 1)).view(-1)
    (want * dlogp).sum().backward()
    t_oracle = time.time() - t0
    want = want.detach()
    logit_rms = rms(lg.detach())
    del lg
    G = o["params"].like(F32)
    tape = {}
    lp = eng.head_forward(x.to(dev), L["sel"], targets.to(dev), tape)
    dx = eng.head_backward(tape, dlogp.to(dev), G)
    e = lp.cpu() - want
    print(f"head on {KN * C} rows x {cfg.vocab}: |logp - oracle| rms {rms(e):.2e} max {float(e.abs().max()):.2e} (logit rms {logit_rms:.2f}); "
          f"oracle fwd+bwd {t_oracle:.1f} s")
    # one bf16-operand dot product of length 3584: error ~ 2^-9 x logit rms
    assert rms(e) <= 1.5 * 2.0 ** -9 * logit_rms and float(e.abs().max()) <= 6 * 2.0 ** -9 * logit_rms     # measured 0.85 / 3.6 x 2^-9 x logit rms
    got = export_state_dict(G)
    r_dx = rel_fro(dx.cpu(), xr.grad)
    print(f"   d stream rel Frobenius err {r_dx:.3e}")
    assert r_dx <= 5e-3                                                   # measured 2.3e-3
    for n in names:
        r = rel_fro(got[n].float().cpu(), w[n].grad)
        print(f"   grad {n:24s} rel Frobenius err {r:.3e}")
        assert r <= 5e-3, (n, r)                                          # measured 2.4e-3
        w[n].requires_grad_(False); w[n].grad = None
    # precise head
    hp = K.norm_pair(x.to(dev), o["params"]["llm.norm_w"], None, cfg.rms_eps)
    lgp = K.gemm_pair(K.gather_rows(hp[0], L["sel"]), K.gather_rows(hp[1], L["sel"]), o["params"]["llm.lm_head"])
    ep = K.logprob_fwd(lgp, targets.to(dev))[0].cpu() - want
    print(f"   precise mode: rms {rms(ep):.2e} max {float(ep.abs().max()):.2e}")
    assert float(ep.abs().max()) <= 1e-4


def test_vision_block_forward_backward_on_the_16_frame_grid(one, dev):
    o = one
    cfg, eng, w, pr = o["cfg"], o["eng"], o["w"], o["prompt"]
    grid = tuple(pr.grids[0])
    assert grid == (8, 20, 26)
    rows, grid_o = O.patchify_frames(o["frames"].cpu(), o["ocfg"])
    assert tuple(grid_o) == grid
    rows = rows.to(torch.bfloat16).float()
    names = ["visual.blocks.0.attn.qkv.weight", "visual.blocks.0.attn.qkv.bias", "visual.blocks.0.attn.proj.weight",
             "visual.blocks.0.mlp.fc1.weight", "visual.blocks.0.mlp.fc2.weight", "visual.blocks.0.norm1.weight", "visual.blocks.0.norm2.bias",
             "visual.merger.mlp.0.weight", "visual.merger.mlp.2.weight", "visual.merger.ln_q.weight", "visual.patch_embed.proj.weight"]
    for n in names:
        w[n].requires_grad_(True)
    g = torch.Generator().manual_seed(13)
    d_out = (torch.randn(grid[0] * grid[1] * grid[2] // 4, cfg.hidden, generator=g) * 0.1).to(torch.bfloat16).float()   # what the engine is handed
    t0 = time.time()
    want = O.vit_forward(w, o["ocfg"], rows, [grid])
    (want * d_out).sum().backward()
    t_oracle = time.time() - t0
    want = want.detach()
    G = o["params"].like(F32)
    tape = {}
    out = eng.vit_forward(pr.pix, [grid], tape)
    eng.vit_backward(tape, d_out.to(dev).to(torch.bfloat16), G)
    e = out.float().cpu() - want
    print(f"vision tower (1 block) on grid {grid}: |out - oracle| rms {rms(e):.2e} max {float(e.abs().max()):.2e}; out rms {rms(want):.2e}; "
          f"oracle fwd+bwd {t_oracle:.1f} s")
    assert rms(e) <= 6e-3 * rms(want) and float(e.abs().max()) <= 1e-2 * float(want.abs().max())      # measured 4.0e-3 rms
    got = export_state_dict(G)
    got["visual.patch_embed.proj.weight"] = got["visual.patch_embed.proj.weight"].reshape(cfg.vit_dim, -1)
    for n in names:
        r = rel_fro(got[n].float().cpu(), w[n].grad)
        print(f"   grad {n:40s} rel Frobenius err {r:.3e}")
        assert r <= 1e-2, (n, r)                                          # measured 3.6e-3 .. 5.7e-3
        w[n].requires_grad_(False); w[n].grad = None
    merged, rev = eng._vit_forward_precise(pr.pix, [grid])
    assert rev is None
    ep = merged.cpu() - want
    print(f"   precise mode: rms {rms(ep):.2e} max {float(ep.abs().max()):.2e}")
    assert rms(ep) <= 1e-4 * rms(want) and float(ep.abs().max()) <= 1e-3 * float(want.abs().max())
