"""GPU parity of the whole forward / backward path against the fp32 CPU oracle on the golden Qwen2.5-VL miniature
(window-attention RMSNorm/SwiGLU vision tower with ragged windows and a zero-padded SwiGLU width,
RMSNorm merger; real head sizes: LLM 128, ViT 80).  Both sides use the SAME bf16-rounded weights; the oracle computes in fp32,
the engine in bf16 activations + fp32 residual stream, so the tolerances below are bf16-activation budgets:
  ViT output 3e-2 abs, per-token log-probs 1e-3 abs on the 2-layer model (north_star budget), parameter
  gradients 4% of each tensor's max + small abs floor."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny25                      # noqa: E402
from oracle import qwen2vl_fp32 as O                   # noqa: E402
from spacer_amd import kernels as K                    # noqa: E402
from spacer_amd.qwen2vl.config import TINY25 as TINY             # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine    # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict  # noqa: E402


@pytest.fixture(scope="module")
def setup(dev):
    g = load_tiny25()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}      # bf16-rounded weights, fp32 container
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    eng = Qwen2VLEngine(TINY, params)
    pix, grid = K.patchify(g["frames"].to(dev), kpad=TINY.patch_kpad)
    rows, grid_o = O.patchify_frames(g["frames"], g["cfg"])
    assert tuple(grid) == tuple(grid_o) == g["grid"]
    return dict(g=g, params=params, wb=wb, eng=eng, pix=pix, rows=rows.to(torch.bfloat16).float(), grid=tuple(grid))


def test_config_matches_fixture(setup):
    a, b = TINY.as_oracle_dict(), setup["g"]["cfg"]
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
    assert err < 8e-3, f"log-prob max abs err {err}"   # bf16 activations; see DESIGN.md "numerics"
    # the reference's own bf16 eager numerics (every op output rounded to bf16, bf16 residual stream and logits),
    # emulated on the CPU: this engine must be at least as close to the fp32 truth as the reference path is
    from oracle import qwen2vl_bf16_emul as E
    ref_bf16 = E.completion_logps(s["wb"], g["cfg"], g["prompt"], g["completions"], s["rows"], [s["grid"]])
    err_ref = (ref_bf16 - want).abs().max()
    err_pair = (lp.cpu() - ref_bf16).abs().max()
    print(f"max |logp - fp32 oracle|: engine {float(err):.2e}, reference-style bf16 eager {float(err_ref):.2e}; "
          f"engine vs bf16 eager {float(err_pair):.2e}")
    assert err <= err_ref + 1e-3
    # text-only prompt (no video) goes through the same path
    lp2 = s["eng"].score_group(g["prompt"][-9:].to(dev), g["completions"].to(dev), None, None)
    want2 = O.completion_logps(s["wb"], g["cfg"], g["prompt"][-9:], g["completions"], None, None)
    assert (lp2.cpu() - want2).abs().max() < 8e-3


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


def test_greedy_rollout_is_oracle_argmax_and_uses_video_seconds(setup):
    """Rollout on the Qwen2.5-VL miniature: greedy tokens are arg-maxes of the oracle's next-token logits (bf16 budget
    3e-2), and the processor's second_per_grid_ts reaches the rollout's M-RoPE positions (the reference keeps it for
    generate, TR:463, and drops it for scoring, TR:519-520)."""
    from spacer_amd.rollout import PromptInput, RolloutEngine, SamplingParams
    s, g = setup, setup["g"]
    dev = s["pix"].device
    roll = RolloutEngine(s["eng"])
    sp = SamplingParams(max_new_tokens=5, top_k=1, top_p=1.0, suppress_eos=True)
    for sec in (None, [2.0]):
        pr = PromptInput(g["prompt"].to(dev), s["pix"], [s["grid"]], second_per_grid_ts=sec)
        out = roll.generate([pr], 2, sp, use_graph=False)
        assert torch.equal(out[0], out[1])
        comp = out[0].cpu()
        ids = torch.cat([g["prompt"], comp])
        ve = O.vit_forward(s["wb"], g["cfg"], s["rows"], [s["grid"]])
        pos3, _ = O.mrope_position_ids(ids.tolist(), [s["grid"]], g["cfg"], second_per_grid_ts=sec)
        lg = O.llm_forward(s["wb"], g["cfg"], O.embed_with_video(s["wb"], g["cfg"], ids, ve), pos3)
        P = g["prompt"].numel()
        for t in range(5):
            row = lg[P - 1 + t].clone()
            row[TINY.eos_token_id] = float("-inf")
            assert float(row.max() - row[comp[t]]) < 3e-2, (sec, t, float(row.max() - row[comp[t]]))
