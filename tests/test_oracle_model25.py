"""CPU: the fp32 oracle's Qwen2.5-VL path (window-attention RMSNorm/SwiGLU vision tower, RMSNorm merger, temporally
scaled M-RoPE) reproduces HF transformers' Qwen2_5_VLForConditionalGeneration on the golden vectors of
scripts/make_golden_model25.py; and the product's host-side planners (window plan, positions, weight layout)
agree with the oracle.  Tolerance: fp32 round-off of a 2-layer model, 5e-5 absolute on logits/log-probs."""
import torch

from golden_util import load_tiny25
from oracle import qwen2vl_fp32 as O
from spacer_amd.qwen2vl import positions as POS
from spacer_amd.qwen2vl.config import QWEN2_5_VL_7B, TINY25
from spacer_amd.qwen2vl.weights import FlatParams, export_state_dict, load_state_dict


def test_vit25_matches_hf():
    g = load_tiny25()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    assert tuple(grid) == g["grid"]
    out = O.vit_forward(g["w"], g["cfg"], rows, [grid])
    assert torch.allclose(out, g["hf_vit"], atol=2e-5, rtol=1e-4)


def test_logits_and_logps_match_hf():
    g = load_tiny25()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    ids = torch.cat([g["prompt"], g["completions"][0]])
    lg = O.full_logits(g["w"], g["cfg"], ids, rows, [grid])
    assert (lg - g["hf_logits_row0"]).abs().max() < 5e-5
    lp = O.completion_logps(g["w"], g["cfg"], g["prompt"], g["completions"], rows, [grid])
    assert (lp - g["hf_logps"]).abs().max() < 5e-5


def test_window_index_and_positions_match_hf():
    g = load_tiny25()
    cfg = g["cfg"]
    win, lens = O.vit_window_index([g["grid"]], cfg)
    assert torch.equal(win, g["hf_window_index"].long())
    assert [0] + torch.tensor(lens).cumsum(0).tolist() == g["hf_cu_window"].tolist()
    ids = torch.cat([g["prompt"], g["completions"][0]]).tolist()
    pos, delta = O.mrope_position_ids(ids, [g["grid"]], cfg)
    assert torch.equal(pos, g["hf_pos"]) and delta == g["hf_delta"]
    pos2, _ = O.mrope_position_ids(ids, [g["grid"]], cfg, second_per_grid_ts=[2.0])
    assert torch.equal(pos2, g["hf_pos_sec2"].long())
    # the 4.x-era rule floors index * seconds * tokens_per_second and advances past the largest temporal position
    thin = [cfg["video_token_id"]] * 4 + [5, 6]
    p5, _ = O.mrope_position_ids(thin, [(4, 2, 2)], cfg, second_per_grid_ts=[1.5])
    p4, _ = O.mrope_position_ids(thin, [(4, 2, 2)], cfg, second_per_grid_ts=[1.5], era_rule=True)
    assert p5[0, :4].tolist() == [0, 2, 4, 6] and int(p5[0, 4]) == 1           # tps * int(1.5) = 2 per step; text resumes at max(h, w)
    assert p4[0, :4].tolist() == [0, 3, 6, 9] and int(p4[0, 4]) == 10          # floor(i * 1.5 * 2); text resumes after 9


def test_product_planners_agree_with_oracle():
    g = load_tiny25()
    cfg = g["cfg"]
    for kcfg, grids in ((TINY25, [g["grid"]]), (TINY25, [(1, 4, 4), (3, 6, 10)]), (QWEN2_5_VL_7B, [(8, 20, 26)])):
        od = kcfg.as_oracle_dict()
        unit, rows, segs = POS.vit_window_plan(grids, kcfg)
        win, lens = O.vit_window_index(grids, od)
        assert torch.equal(unit, win) and [s[1] for s in segs] == lens
        assert [s[0] for s in segs] == ([0] + torch.tensor(lens).cumsum(0).tolist())[:-1]
        mu = kcfg.merge ** 2
        assert torch.equal(rows.view(-1, mu)[:, 0], unit * mu) and rows.numel() == sum(t * h * w for t, h, w in grids)
    ids = torch.cat([g["prompt"], g["completions"][0]]).tolist()
    for era in (False, True):
        for sec in (None, [2.0], [1.5]):
            a, da = POS.mrope_positions(ids, [g["grid"]], TINY25, era, sec)
            b, db = O.mrope_position_ids(ids, [g["grid"]], cfg, era_rule=era, second_per_grid_ts=sec)
            assert torch.equal(a, b) and da == db, (era, sec)


def test_weight_layout_round_trip():
    """Checkpoint names -> fused / zero-padded engine layout -> checkpoint names is the identity (fp16-representable values)."""
    g = load_tiny25()
    params = FlatParams.empty(TINY25, "cpu", dtype=torch.float32)
    load_state_dict(params, g["w"])
    back = export_state_dict(params)
    back["visual.patch_embed.proj.weight"] = back["visual.patch_embed.proj.weight"].reshape(TINY25.vit_dim, -1)
    assert set(back) == set(g["w"])
    for k, v in g["w"].items():
        assert torch.equal(back[k], v), k
    I, Ip = TINY25.vit_mlp, TINY25.vit_mlp_pad
    assert Ip == 448 and float(params["vit.0.gu_w"][I:Ip].abs().sum()) == 0 and float(params["vit.0.down_w"][:, I:].abs().sum()) == 0
