"""GPU: oracle parity at REAL DEPTH -- the full Qwen2-VL-2B architecture (BASELINE.json configs[0]/[1]: 28 decoder layers,
hidden 1536, 12/2 heads of 128, intermediate 8960, vocab 151936, lm_head TIED to the embedding table, ViT 32 x 1280),
seeded random-init bf16 weights, one prompt of 242 tokens (4 frames 112x140 -> 40 video tokens + 200 text), K = 2
completions of 24 tokens.  The fp32 oracle (oracle/qwen2vl_fp32.py) runs the same weights on the GPU box's host cores
(forward ~5 s, autograd backward ~10 s) -- this is the quantity SG_RLVR_trainer.py:353-366 produces and the north-star pins.

Tolerances (DESIGN.md section 4 has the per-operator table they come from):
  * log-probs: the bf16-operand floor at 28 layers is rms ~1e-2 / max ~2-3e-2 (CPU emulation of the engine's rounding points,
    oracle/qwen2vl_engine_emul.py; scripts/logp_error_budget.py 2b: 8.7e-3 / 1.9e-2 on its sample); the reference's own
    bf16-eager numerics give rms 2.7e-2 / max 5.7e-2.  Asserted for the FAST path: (i) LAYER BY LAYER, the rms error of the fp32
    residual stream entering each of the 28 decoder layers <= 1.1x the emulation's at the same layer (a kernel that loses more
    than its operand rounding shows at the layer where it happens); (ii) the log-prob rms error <= 1.5x the emulation's on the
    same tokens, absolute caps rms <= 2e-2, max <= 5e-2.  The north-star's 1e-3 is held by the PRECISE scoring mode (hi+lo
    operand pairs on every matmul operand incl. attention, csrc/precise.hip): tests/test_precise_gpu.py asserts it on this
    same model (measured 3e-5).
  * gradients of selected tensors (first / middle / last decoder layer, the tied embedding table, final norm, first / last
    vision block, merger): relative Frobenius error <= 6 % against oracle autograd (2B), <= 10 % at 7B depth.
Round 3: the same test also runs on the headline model itself, Qwen2-VL-7B at full depth (8.29 B parameters; oracle forward +
autograd backward on the host in ~20 s): per-layer error ratio 1.00, log-probs rms 4.1e-2 (emulation 3.8e-2), gradients 3.3-6.6 %.
"""
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import qwen2vl_fp32 as O                                  # noqa: E402
from spacer_amd.qwen2vl.config import QWEN2_VL_2B, QWEN2_VL_7B        # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine                   # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, random_init_   # noqa: E402
from spacer_amd.synthetic import make_prompt                          # noqa: E402

GRAD_NAMES = ["model.layers.27.mlp.down_proj.weight", "model.layers.13.self_attn.q_proj.weight",
              "model.layers.13.self_attn.k_proj.bias", "model.layers.0.mlp.gate_proj.weight", "model.embed_tokens.weight",
              "model.norm.weight", "model.layers.5.input_layernorm.weight", "visual.blocks.31.mlp.fc2.weight",
              "visual.blocks.0.attn.qkv.weight", "visual.merger.mlp.2.weight", "visual.patch_embed.proj.weight"]


@pytest.fixture(scope="module", params=["2b", "7b"])
def depth(dev, request):
    """2b: Qwen2-VL-2B (tied lm_head); 7b: the headline model Qwen2-VL-7B (8.29 B parameters, untied lm_head) -- both at full depth."""
    cfg = QWEN2_VL_2B if request.param == "2b" else QWEN2_VL_7B
    torch.cuda.empty_cache()
    params = FlatParams.empty(cfg, dev)
    random_init_(params, seed=1234)
    eng = Qwen2VLEngine(cfg, params)
    prompt, frames = make_prompt(cfg, 5, 4, 112, 140, 200, dev)
    comps = torch.randint(1000, 150000, (2, 24), generator=torch.Generator().manual_seed(9)).to(dev)
    # the oracle's copy: bf16 values in fp32 containers, original checkpoint names
    w = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    w["visual.patch_embed.proj.weight"] = w["visual.patch_embed.proj.weight"].reshape(cfg.vit_dim, -1)
    assert ("lm_head.weight" not in w) == cfg.tie_embeddings
    ocfg = cfg.as_oracle_dict()
    rows, grid = O.patchify_frames(frames.cpu(), ocfg)
    assert tuple(grid) == tuple(prompt.grids[0])
    yield dict(cfg=cfg, ocfg=ocfg, params=params, eng=eng, prompt=prompt, comps=comps, w=w,
               rows=rows.to(torch.bfloat16).float(), grid=tuple(grid))
    del eng, params
    torch.cuda.empty_cache()


def test_logps_and_gradients_match_oracle_at_full_depth(depth):
    d = depth
    names = GRAD_NAMES + ([] if d["cfg"].tie_embeddings else ["lm_head.weight"])
    eng, pr, comps, params = d["eng"], d["prompt"], d["comps"], d["params"]
    torch.set_num_threads(min(32, torch.get_num_threads()))
    Kn, C = comps.shape
    dlogp = torch.randn(Kn, C, generator=torch.Generator().manual_seed(5)) * 0.5
    # ---- oracle on the host: forward + autograd
    for n in names:
        d["w"][n].requires_grad_(True)
    t0 = time.time()
    want = O.completion_logps(d["w"], d["ocfg"], pr.ids.cpu(), comps.cpu(), d["rows"], [d["grid"]])
    (want * dlogp).sum().backward()
    t_oracle = time.time() - t0
    want = want.detach()
    from oracle import qwen2vl_engine_emul as E
    with torch.no_grad():
        emu = E.completion_logps(d["w"], d["ocfg"], pr.ids.cpu(), comps.cpu(), d["rows"], [d["grid"]])
    e_emu = emu - want
    rms_emu = float(e_emu.pow(2).mean().sqrt())
    # ---- engine
    G = params.like(torch.float32)
    tape = {}
    lp = eng.score_group(pr.ids, comps, pr.pix, pr.grids, tape=tape)
    eng_x = [t["x_in"].cpu() for t in tape["llm"]]             # the fp32 residual stream entering every decoder layer
    eng.backward_group(tape, dlogp.to(comps.device), G)
    # ---- per-layer bound: engine vs the CPU emulation of its own rounding points, both against the fp32 stream, LAYER BY LAYER on
    # the packed group layout.  A kernel that loses more than its operand rounding shows up at the layer where it happens -- before
    # depth amplification blurs it -- as rms(engine - fp32) > rms(emulation - fp32).
    from oracle import cpu_path as CP
    with torch.no_grad():
        P = pr.ids.numel()
        ids = torch.cat([pr.ids.cpu(), comps.cpu().reshape(-1)])
        pos3, delta = O.mrope_position_ids(pr.ids.cpu().tolist(), [d["grid"]], d["ocfg"])
        pos = torch.cat([pos3] + [(P + delta + torch.arange(C)).view(1, C).expand(3, C)] * Kn, 1)
        mask = CP.group_mask(P, Kn, C)
        streams = {}
        for tag, R in (("fp32", E.Rounder((), ())), ("emu", E.Rounder())):
            ve = E.vit_forward(d["w"], d["ocfg"], d["rows"], [d["grid"]], R)
            col = []
            E.llm_hidden(d["w"], d["ocfg"], O.embed_with_video(d["w"], d["ocfg"], ids, ve), pos, R, mask, collect=col)
            streams[tag] = [c["x_in"] for c in col]
    rmsf = lambda t: float(t.double().pow(2).mean().sqrt())                              # noqa: E731
    ratios = []
    for i, (xe, xm, xf) in enumerate(zip(eng_x, streams["emu"], streams["fp32"])):
        r_eng, r_emu = rmsf(xe - xf), rmsf(xm - xf)
        ratios.append(r_eng / max(r_emu, 1e-12))
        if i in (0, 1, 2, 7, 14, 21, 27):
            print(f"   layer {i:2d} input stream: rms err engine {r_eng:.3e}  emulation {r_emu:.3e}  ratio {ratios[-1]:.2f}  (stream rms {rmsf(xf):.2e})")
    print(f"   per-layer engine / emulation error ratio: max {max(ratios):.2f} at layer {ratios.index(max(ratios))}, mean {sum(ratios) / len(ratios):.2f}")
    assert max(ratios) <= 1.1, ratios                     # measured 1.00 at every layer
    err = lp.cpu() - want
    rms, mx = float(err.pow(2).mean().sqrt()), float(err.abs().max())
    print(f"{'Qwen2-VL-2B' if d['cfg'].tie_embeddings else 'Qwen2-VL-7B'} depth: |logp - fp32 oracle| rms {rms:.2e} max {mx:.2e} over {err.numel()} tokens "
          f"(logp range [{float(want.min()):.2f}, {float(want.max()):.2f}]); emulated rounding points rms {rms_emu:.2e} max "
          f"{float(e_emu.abs().max()):.2e}; oracle fwd+bwd {t_oracle:.1f} s on the host")
    assert torch.isfinite(lp).all()
    assert rms <= 1.5 * rms_emu + 1e-3, (rms, rms_emu)
    cap_rms, cap_max = (2e-2, 5e-2) if d["cfg"].tie_embeddings else (6e-2, 2e-1)      # 7B (width 3584) amplifies ~4x more than 2B
    assert rms <= cap_rms and mx <= cap_max, (rms, mx)
    got = export_state_dict(G)
    got["visual.patch_embed.proj.weight"] = got["visual.patch_embed.proj.weight"].reshape(d["cfg"].vit_dim, -1)
    bad = []
    for n in names:
        gr, ge = d["w"][n].grad, got[n].float().cpu()
        rel = float((ge - gr).norm() / (gr.norm() + 1e-30))
        print(f"   grad {n:48s} rel Frobenius err {rel:.3e}   |g| {float(gr.norm()):.3e}")
        # the backward linearises around a forward whose residual stream already carries the bf16-operand noise amplified over 28
        # layers (relative stream error at the last layer: 1.2 % at 2B, 3.0 % at 7B -- printed above, equal to the emulation's):
        # 2B measured 1.3-2.1 %, 7B 3.3-6.6 %
        if not rel <= (6e-2 if d["cfg"].tie_embeddings else 1e-1):
            bad.append((n, rel))
    assert not bad, bad
