"""GPU parity of the whole forward / backward path against the fp32 CPU oracle on the golden tiny models
(real head sizes: LLM 128, ViT 80; "tiny" = untied lm_head as Qwen2-VL-7B, "tiny_tied" = lm_head tied to the embedding
table as Qwen2-VL-2B).  Both sides use the SAME bf16-rounded weights; the oracle computes in fp32, the engine with bf16
MFMA operands, fp32 accumulation, an fp32 residual stream and fp32 logits.  Tolerances:
  * per-token log-probs (SG_RLVR_trainer.py:353-366): the north-star's 1e-3 is below what ANY bf16-operand pipeline can
    reach (per-operator budget from oracle/qwen2vl_engine_emul.py -- rms 0.8e-3 / max 2.5e-3 on this
    2-layer model, rms 9e-3 at 28 layers).  What is asserted: the engine sits AT that floor -- rms error over 256 tokens
    <= 1.3x the CPU emulation's rms error, max error <= 5e-3 -- and is far inside the reference's own bf16-eager error;
  * ViT output 3e-2 abs; parameter gradients 4 % of each tensor's max + small abs floor."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny                      # noqa: E402
from oracle import qwen2vl_fp32 as O                   # noqa: E402
from spacer_amd import kernels as K                    # noqa: E402
from spacer_amd.qwen2vl.config import TINY as TINY_UNTIED, TINY_TIED    # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine    # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict  # noqa: E402


TINY = TINY_UNTIED     # shapes shared by both fixtures (they differ in tie_embeddings only)


@pytest.fixture(scope="module", params=["tiny", "tiny_tied"])
def setup(dev, request):
    tied = request.param == "tiny_tied"
    g = load_tiny("tiny_tied_model.npz" if tied else "tiny_model.npz")
    TINY = TINY_TIED if tied else TINY_UNTIED
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}      # bf16-rounded weights, fp32 container
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    eng = Qwen2VLEngine(TINY, params)
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    rows, grid_o = O.patchify_frames(g["frames"], g["cfg"])
    assert tuple(grid) == tuple(grid_o) == g["grid"]
    return dict(g=g, params=params, wb=wb, eng=eng, pix=pix, rows=rows.to(torch.bfloat16).float(), grid=tuple(grid), cfg=TINY)


def test_config_matches_fixture(setup):
    a, b = setup["cfg"].as_oracle_dict(), setup["g"]["cfg"]
    for k in b:
        assert a[k] == b[k] or tuple(a[k]) == tuple(b[k]), k


def test_vit_forward(setup):
    s = setup
    got = s["eng"].vit_forward(s["pix"], [s["grid"]])
    want = O.vit_forward(s["wb"], s["g"]["cfg"], s["rows"], [s["grid"]])
    err = (got.float().cpu() - want).abs().max()
    assert err < 3e-2, err


def test_logps_match_oracle(setup):
    s, g = setup, setup["g"]
    dev = s["pix"].device
    lp = s["eng"].score_group(g["prompt"].to(dev), g["completions"].to(dev), s["pix"], [s["grid"]])
    want = O.completion_logps(s["wb"], g["cfg"], g["prompt"], g["completions"], s["rows"], [s["grid"]])
    err = (lp.cpu() - want).abs().max()
    assert err < 5e-3, f"log-prob max abs err {err}"   # bf16-operand floor of this model: DESIGN.md section 4
    # the reference's own bf16 eager numerics (every op output rounded to bf16, bf16 residual stream and logits),
    # emulated on the CPU: this engine must be at least as close to the fp32 truth as the reference path is
    from oracle import qwen2vl_bf16_emul as E
    ref_bf16 = E.completion_logps(s["wb"], g["cfg"], g["prompt"], g["completions"], s["rows"], [s["grid"]])
    err_ref = (ref_bf16 - want).abs().max()
    err_pair = (lp.cpu() - ref_bf16).abs().max()
    print(f"max |logp - fp32 oracle|: engine {float(err):.2e}, reference-style bf16 eager {float(err_ref):.2e}; "
          f"engine vs bf16 eager {float(err_pair):.2e}")
    assert err <= 0.5 * err_ref
    # text-only prompt (no video) goes through the same path
    lp2 = s["eng"].score_group(g["prompt"][-9:].to(dev), g["completions"].to(dev), None, None)
    want2 = O.completion_logps(s["wb"], g["cfg"], g["prompt"][-9:], g["completions"], None, None)
    assert (lp2.cpu() - want2).abs().max() < 5e-3


def test_logps_sit_at_the_bf16_operand_floor(setup):
    """256 completion tokens (8 rollouts x 32): the engine's error against the fp32 oracle vs the error of the CPU emulation
    of its own rounding points (oracle/qwen2vl_engine_emul.py).  A kernel that loses more than its operand rounding
    (a bf16 accumulator, a bf16 residual, a low-precision exp) shows up as rms(engine) > rms(emulation)."""
    from oracle import qwen2vl_engine_emul as E
    s, g = setup, setup["g"]
    dev = s["pix"].device
    comps = torch.randint(5, 990, (8, 32), generator=torch.Generator().manual_seed(21))
    want = O.completion_logps(s["wb"], g["cfg"], g["prompt"], comps, s["rows"], [s["grid"]])
    emu = E.completion_logps(s["wb"], g["cfg"], g["prompt"], comps, s["rows"], [s["grid"]])
    lp = s["eng"].score_group(g["prompt"].to(dev), comps.to(dev), s["pix"], [s["grid"]]).cpu()
    rms = lambda d: float(d.pow(2).mean().sqrt())                                         # noqa: E731
    e_eng, e_emu = lp - want, emu - want
    print(f"log-prob error vs fp32 oracle over {comps.numel()} tokens: engine rms {rms(e_eng):.2e} max {float(e_eng.abs().max()):.2e}; "
          f"emulated rounding points rms {rms(e_emu):.2e} max {float(e_emu.abs().max()):.2e}; engine vs emulation rms {rms(lp - emu):.2e}")
    assert rms(e_eng) <= 1.3 * rms(e_emu) + 1e-4
    assert float(e_eng.abs().max()) <= 5e-3
    assert rms(e_eng) <= 1.5e-3


def test_sampled_placeholder_ids_are_plain_tokens(setup):
    """A random-init policy does sample <|video_pad|> / <|image_pad|> as completion tokens.  They are ordinary tokens there
    (own embedding row; only the prompt's placeholders take ViT rows): forward, backward and the ViT-gradient row count stay
    consistent, and the log-probs before the injected position are untouched (causality)."""
    s, g = setup, setup["g"]
    dev = s["pix"].device
    cfg = s["cfg"]
    comps = g["completions"].clone()
    base = s["eng"].score_group(g["prompt"].to(dev), comps.to(dev), s["pix"], [s["grid"]])
    comps[1, 3] = cfg.video_token_id
    comps[2, 0] = cfg.image_token_id
    tape = {}
    lp = s["eng"].score_group(g["prompt"].to(dev), comps.to(dev), s["pix"], [s["grid"]], tape=tape)
    assert torch.isfinite(lp).all()
    assert torch.equal(lp[0], base[0]) and torch.equal(lp[1, :3], base[1, :3])
    G = s["params"].like(torch.float32)
    s["eng"].backward_group(tape, torch.ones_like(lp), G)
    assert torch.isfinite(G.flat).all() and float(G["vit.patch_w"].abs().max()) > 0


def test_several_groups_per_pass_equal_group_by_group(setup):
    """Qwen2VLEngine.score_groups / GRPOEngine.score_and_backward_multi: two prompt groups (different prompts, one with a
    shorter text tail) in ONE token-packed pass give the log-probs of the two single-group passes, and the accumulated
    gradient of the pair equals the sum of the two single-group backward passes (same loss weighting)."""
    from spacer_amd.grpo import GRPOEngine, GRPOHyper
    from spacer_amd.rollout import PromptInput
    s, g = setup, setup["g"]
    dev = s["pix"].device
    cfg = s["cfg"]
    p0 = PromptInput(g["prompt"].to(dev), s["pix"], [s["grid"]])
    p1 = PromptInput(g["prompt"][:-4].to(dev), s["pix"], [s["grid"]])
    gen = torch.Generator().manual_seed(33)
    c0, c1 = torch.randint(5, 990, (3, 6), generator=gen).to(dev), torch.randint(5, 990, (3, 6), generator=gen).to(dev)
    eng = s["eng"]
    both = eng.score_groups([(p0.ids, p0.pix, p0.grids), (p1.ids, p1.pix, p1.grids)], [c0, c1])
    one0, one1 = eng.score_group(p0.ids, c0, p0.pix, p0.grids), eng.score_group(p1.ids, c1, p1.pix, p1.grids)
    assert float((both[:3] - one0).abs().max()) < 2e-3 and float((both[3:] - one1).abs().max()) < 2e-3
    adv = [torch.tensor([1.0, -0.5, 0.2]), torch.tensor([-1.0, 0.3, 0.9])]
    ge_a = GRPOEngine(cfg, s["params"], GRPOHyper(num_generations=3), ref=s["params"])
    ge_a.score_and_backward(p0, c0, adv[0].to(dev), grad_scale=0.5)
    ge_a.score_and_backward(p1, c1, adv[1].to(dev), grad_scale=0.5)
    ge_b = GRPOEngine(cfg, s["params"], GRPOHyper(num_generations=3), ref=s["params"])
    res = ge_b.score_and_backward_multi([p0, p1], [c0, c1], adv, grad_scale=0.5)
    assert tuple(res["logps"].shape) == (6, 6)
    a, b = ge_a.G.flat, ge_b.G.flat
    assert float((a - b).abs().max()) <= 2e-2 * float(a.abs().max()) + 1e-6, (float((a - b).abs().max()), float(a.abs().max()))


def test_backward_matches_oracle_autograd(setup):
    s, g = setup, setup["g"]
    dev = s["pix"].device
    eng, params = s["eng"], s["params"]
    Kn, C = g["completions"].shape
    dlogp = (torch.randn(Kn, C, generator=torch.Generator().manual_seed(5)) * 0.5)
    # oracle gradients
    wr = {k: v.clone().requires_grad_(True) for k, v in s["wb"].items()}
    lp_o = O.completion_logps(wr, g["cfg"], g["prompt"], g["completions"], s["rows"], [s["grid"]])
    (lp_o * dlogp).sum().backward()
    # engine gradients
    G = params.like(torch.float32)
    tape = {}
    eng.score_group(g["prompt"].to(dev), g["completions"].to(dev), s["pix"], [s["grid"]], tape=tape)
    eng.backward_group(tape, dlogp.to(dev), G)
    got = export_state_dict(G)
    got["visual.patch_embed.proj.weight"] = got["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    assert ("lm_head.weight" in wr) != s["cfg"].tie_embeddings       # tied: the table's gradient sums both of its uses
    worst = []
    for name, ref in wr.items():
        gr = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        ge = got[name].float().cpu()
        scale = float(gr.abs().max())
        err = float((ge - gr).abs().max())
        worst.append((err / (scale + 1e-6), name, err, scale, err <= 0.04 * scale + 2e-4))
    bad = [w for w in worst if not w[4]]
    worst.sort(reverse=True)
    print("worst relative grad errors:")
    for w in worst[:8]:
        print("   %.3f %-50s err %.3e max|g| %.3e" % w[:4])
    assert not bad, "gradient mismatch:\n" + "\n".join("%-50s err %.3e max|g| %.3e" % (b[1], b[2], b[3]) for b in bad)


def test_sft_loss_and_gradients_match_oracle(setup):
    """The supervised objective of open_r1/sft.py (HF causal-LM loss, pad and visual tokens ignored) through
    score_sequence / sft_forward_backward: loss within 5e-3 of the fp32 oracle, gradients within 4 % of each tensor's max."""
    from spacer_amd.grpo import GRPOEngine, GRPOHyper
    from spacer_amd.open_r1.sft import label_mask
    s, g = setup, setup["g"]
    dev = s["pix"].device
    ids = torch.cat([g["prompt"], g["completions"][0], g["completions"][1]])
    keep = label_mask(ids, TINY.pad_token_id, (TINY.vision_start_id, TINY.vision_end_id, TINY.video_token_id, TINY.image_token_id))
    assert 0 < int(keep.sum()) < ids.numel()
    # oracle: mean CE of the shifted labels
    wr = {k: v.clone().requires_grad_(True) for k, v in s["wb"].items()}
    lg = O.full_logits(wr, g["cfg"], ids, s["rows"], [s["grid"]])
    lp = torch.log_softmax(lg[:-1], -1).gather(1, ids[1:, None]).squeeze(1)
    m = keep[1:].float()
    loss_o = -(lp * m).sum() / m.sum()
    loss_o.backward()
    ge = GRPOEngine(s["cfg"], s["params"], GRPOHyper(), ref=s["params"])
    loss = ge.sft_forward_backward(ids.to(dev), s["pix"], [s["grid"]], keep)
    assert abs(loss - float(loss_o.detach())) < 5e-3, (loss, float(loss_o.detach()))
    got = export_state_dict(ge.G)
    got["visual.patch_embed.proj.weight"] = got["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    bad = []
    for name, ref in wr.items():
        gr = ref.grad if ref.grad is not None else torch.zeros_like(ref)
        err, scale = float((got[name].float().cpu() - gr).abs().max()), float(gr.abs().max())
        if err > 0.04 * scale + 2e-4:
            bad.append((name, err, scale))
    assert not bad, bad


def test_ready_sequence_does_not_depend_on_the_data(setup):
    """Data-parallel overlap (grpo.GradReducer): every rank must issue the same sequence of collectives.  The parameter ranges
    reported as final during the last backward are therefore the same for a prompt WITH vision input and a text-only one (SFT
    'messages' rows, non-image / video rows): a text-only rank still reports the merger / vision ranges, which are final."""
    s, g = setup, setup["g"]
    dev = s["pix"].device
    seqs = []
    for pix, grids, prompt in ((s["pix"], [s["grid"]], g["prompt"]), (None, None, g["prompt"][-9:])):
        G = s["params"].like(torch.float32)
        tape, seen = {}, []
        lp = s["eng"].score_group(prompt.to(dev), g["completions"].to(dev), pix, grids, tape=tape)
        s["eng"].backward_group(tape, torch.ones_like(lp), G, on_ready=seen.append)
        seqs.append(seen)
    assert seqs[0] == seqs[1] and "vit.patch_w" in seqs[1] and "merger." in seqs[1]
    assert len(set(seqs[0])) == len(seqs[0])                      # every range reported exactly once
