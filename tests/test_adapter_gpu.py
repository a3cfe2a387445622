"""GPU: the two model call shapes the reference trainer relies on, through spacer_amd.hf_adapter.SpacerModel:
``model.generate(**prompt_inputs, generation_config=...)`` (TR:463) and ``model(input_ids, **kw).logits`` (TR:357) -- the second
one driven by a restatement of the reference's own ``_get_per_token_logps`` (TR:353-366)."""
from types import SimpleNamespace

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load_tiny                                         # noqa: E402
from oracle import qwen2vl_fp32 as O                                      # noqa: E402
from spacer_amd.hf_adapter import SpacerModel                             # noqa: E402
from spacer_amd.qwen2vl.config import TINY                                # noqa: E402
from spacer_amd.qwen2vl.engine import Qwen2VLEngine                       # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict   # noqa: E402


@pytest.fixture(scope="module")
def model(dev):
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    # what the HF processor hands the trainer (TR:417-425): fp32 pixel rows, grid, ids with left padding + mask
    prompt_inputs = dict(input_ids=g["prompt"].view(1, -1), attention_mask=torch.ones(1, g["prompt"].numel(), dtype=torch.long),
                         pixel_values_videos=rows, video_grid_thw=torch.tensor([list(grid)]))
    wb = {k: v.float().cpu() for k, v in export_state_dict(params).items()}
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(TINY.vit_dim, -1)
    return dict(g=g, m=SpacerModel(Qwen2VLEngine(TINY, params), seed=3), inputs=prompt_inputs, rows=rows, grid=tuple(grid), wb=wb)


def test_generate_call_shape(model):
    g, m = model["g"], model["m"]
    P = g["prompt"].numel()
    gc = SimpleNamespace(max_new_tokens=7, do_sample=True, top_p=0.95, temperature=1.0, num_return_sequences=4, pad_token_id=TINY.pad_token_id)
    out = m.generate(**model["inputs"], generation_config=gc)                    # TR:463
    assert out.dtype == torch.int64 and tuple(out.shape) == (4, P + 7)
    assert torch.equal(out[:, :P].cpu(), g["prompt"].view(1, -1).expand(4, -1))   # TR:465: prompt_ids = out[:, :prompt_length]
    comp = out[:, P:]
    assert int(comp.min()) >= 0 and int(comp.max()) < TINY.vocab and len({tuple(r.tolist()) for r in comp}) > 1
    # greedy: do_sample False -> the K copies coincide and equal the oracle's arg-max continuation
    gc2 = SimpleNamespace(max_new_tokens=3, do_sample=False, num_return_sequences=2, pad_token_id=TINY.pad_token_id)
    out2 = m.generate(**model["inputs"], generation_config=gc2)
    assert torch.equal(out2[0], out2[1])
    lg = O.full_logits(model["wb"], g["cfg"], g["prompt"], model["rows"].to(torch.bfloat16).float(), [model["grid"]])
    top2 = lg[-1].topk(2).values
    if float(top2[0] - top2[1]) > 2e-2:                                          # unambiguous arg-max only
        assert int(out2[0, P]) == int(lg[-1].argmax())


def _get_per_token_logps(model, input_ids, **kwargs):
    """TR:353-366 restated: full logits, shift, per-row log_softmax + gather."""
    logits = model(input_ids, **kwargs).logits
    logits = logits[:, :-1, :]
    input_ids = input_ids[:, 1:]
    out = []
    for logits_row, ids_row in zip(logits, input_ids):
        lp = logits_row.log_softmax(dim=-1)
        out.append(torch.gather(lp, dim=1, index=ids_row.unsqueeze(1)).squeeze(1))
    return torch.stack(out)


def test_forward_logits_call_shape(model):
    g, m = model["g"], model["m"]
    P, (Kn, C) = g["prompt"].numel(), g["completions"].shape
    ids = torch.stack([torch.cat([g["prompt"], c]) for c in g["completions"]]).to(m.device)
    # TR:517-518: pixel rows and grid repeated once per sequence
    kw = dict(pixel_values_videos=model["rows"].repeat(Kn, 1), video_grid_thw=torch.tensor([list(model["grid"])] * Kn))
    lp = _get_per_token_logps(m, ids, **kw)[:, P - 1:]                           # TR:528
    assert tuple(lp.shape) == (Kn, C)
    want = O.completion_logps(model["wb"], g["cfg"], g["prompt"], g["completions"], model["rows"].to(torch.bfloat16).float(), [model["grid"]])
    assert float((lp.cpu() - want).abs().max()) < 5e-3                           # bf16-operand floor of this model
    # and it is the same quantity the engine's shared-prompt scoring produces
    eng_lp = m.engine.score_group(g["prompt"].to(m.device), g["completions"].to(m.device), m._pixels(model["inputs"])[0], [model["grid"]])
    assert float((lp - eng_lp).abs().max()) < 5e-3


def test_forward_with_a_sampled_placeholder_id_in_the_completion(model):
    """A random-init policy does sample <|video_pad|> as a completion token.  Through the adapter it is an ordinary token (own
    embedding row, text position), exactly as in the engine's shared-prompt scoring: no exception, no extra vision row consumed,
    and the log-probs before the injected position are untouched (causality)."""
    g, m = model["g"], model["m"]
    P = g["prompt"].numel()
    comps = g["completions"].clone()
    base_ids = torch.stack([torch.cat([g["prompt"], c]) for c in comps]).to(m.device)
    comps[1, 3] = TINY.video_token_id
    ids = torch.stack([torch.cat([g["prompt"], c]) for c in comps]).to(m.device)
    kw = dict(pixel_values_videos=model["rows"], video_grid_thw=torch.tensor([list(model["grid"])]))
    lp0 = _get_per_token_logps(m, base_ids, **kw)[:, P - 1:]
    lp1 = _get_per_token_logps(m, ids, **kw)[:, P - 1:]
    assert torch.isfinite(lp1).all()
    assert torch.equal(lp0[0], lp1[0]) and torch.equal(lp0[1, :3], lp1[1, :3]) and not torch.equal(lp0[1, 4:], lp1[1, 4:])
    eng_lp = m.engine.score_group(g["prompt"].to(m.device), comps.to(m.device), m._pixels(model["inputs"])[0], [model["grid"]])
    assert float((lp1 - eng_lp).abs().max()) < 5e-3


def test_precise_adapter_logits_hold_the_north_star_tolerance(model):
    """``SpacerModel(engine, precise=True)``: the reference's own ``_get_per_token_logps`` (TR:353-366) run on the adapter's logits
    lands within 1e-4 of the fp32 oracle on the miniature (the fast adapter: ~3e-3)."""
    g, m = model["g"], model["m"]
    pm = SpacerModel(m.engine, m.roll, precise=True)
    P, (Kn, C) = g["prompt"].numel(), g["completions"].shape
    ids = torch.stack([torch.cat([g["prompt"], c]) for c in g["completions"]]).to(m.device)
    kw = dict(pixel_values_videos=model["rows"], video_grid_thw=torch.tensor([list(model["grid"])]))
    lp = _get_per_token_logps(pm, ids, **kw)[:, P - 1:]
    want = O.completion_logps(model["wb"], g["cfg"], g["prompt"], g["completions"], model["rows"].to(torch.bfloat16).float(), [model["grid"]])
    assert float((lp.cpu() - want).abs().max()) < 1e-4
