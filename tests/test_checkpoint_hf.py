"""CPU: a checkpoint written by this repo loads in HF transformers (what SpaceR-Eval does with the trained artefact,
EV/data_utils/vsibench.py:79-93) and computes the golden logits; the engine's config round-trips through config.json.
Covers both families (Qwen2-VL and Qwen2.5-VL miniatures).  Skipped when transformers is not importable."""
import json
import os

import pytest
import torch

from golden_util import load_tiny, load_tiny25
from oracle import qwen2vl_fp32 as O
from spacer_amd.qwen2vl.checkpoint import config_from_hf, config_of_dir, hf_config_dict, read_checkpoint, write_checkpoint
from spacer_amd.qwen2vl.config import QWEN2_5_VL_7B, QWEN2_VL_2B, QWEN2_VL_7B, TINY, TINY25
from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict


@pytest.mark.parametrize("cfg", [TINY, TINY25, QWEN2_VL_7B, QWEN2_VL_2B, QWEN2_5_VL_7B])
def test_config_round_trip(cfg):
    assert config_from_hf(json.loads(json.dumps(hf_config_dict(cfg)))) == cfg


class _FakeProcessor:
    def save_pretrained(self, d):
        with open(os.path.join(d, "preprocessor_config.json"), "w") as f:
            f.write("{}")


@pytest.mark.parametrize("family", ["qwen2", "qwen2_5"])
def test_written_checkpoint_loads_in_hf_and_matches_golden_logits(tmp_path, family):
    transformers = pytest.importorskip("transformers")
    cfg, g = (TINY, load_tiny()) if family == "qwen2" else (TINY25, load_tiny25())
    params = FlatParams.empty(cfg, "cpu", dtype=torch.float32)
    load_state_dict(params, g["w"])
    out = str(tmp_path / "ckpt")
    write_checkpoint(out, params, processor=_FakeProcessor(), extra_state={"global_step": 3})
    assert sorted(os.listdir(out)) == ["config.json", "model.safetensors", "preprocessor_config.json", "trainer_state.json"]
    assert config_of_dir(out) == cfg
    back = read_checkpoint(out)
    assert all(v.dtype == torch.bfloat16 for v in back.values())
    # a second generation: written from the first one's directory, config.json passes through untouched
    out2 = str(tmp_path / "ckpt2")
    write_checkpoint(out2, params, source_dir=out)
    assert open(os.path.join(out, "config.json")).read() == open(os.path.join(out2, "config.json")).read()
    assert os.path.exists(os.path.join(out2, "preprocessor_config.json"))
    m = transformers.AutoModelForImageTextToText.from_pretrained(out, dtype=torch.float32, attn_implementation="eager").eval()
    rows, grid = O.patchify_frames(g["frames"], g["cfg"])
    ids = torch.cat([g["prompt"], g["completions"][0]])[None]
    mm = ((ids == cfg.video_token_id) * 2 + (ids == cfg.image_token_id)).int()
    with torch.no_grad():
        lg = m(input_ids=ids, pixel_values_videos=rows, video_grid_thw=torch.tensor([grid]), mm_token_type_ids=mm).logits[0]
    # golden weights are fp16-representable but not all bf16-representable: compare against the oracle on the SAME
    # bf16-rounded weights (HF vs fp32 oracle round-off), and loosely against the golden HF logits
    wb = {k: v.float() for k, v in back.items()}
    wb["visual.patch_embed.proj.weight"] = wb["visual.patch_embed.proj.weight"].reshape(cfg.vit_dim, -1)
    want = O.full_logits(wb, g["cfg"], ids[0], rows, [grid])
    assert (lg - want).abs().max() < 5e-5
    assert (lg - g["hf_logits_row0"]).abs().max() < 5e-2


def test_reader_ignores_trainer_side_files(tmp_path):
    """An HF Trainer output directory (the reference's SFT checkpoint, a checkpoint-N of a GRPO run) also holds
    training_args.bin / optimizer.pt / scheduler.pt / rng_state.pth: only the weight shards are read; an index file decides
    when there is one; a name that is neither a directory nor a resolvable hub snapshot fails with a clear error."""
    import json as _json

    import pytest as _pytest
    from safetensors.torch import save_file

    from spacer_amd.qwen2vl.checkpoint import read_checkpoint
    d = tmp_path / "ck"
    d.mkdir()
    save_file({"model.language_model.norm.weight": torch.ones(4)}, str(d / "model-00001-of-00002.safetensors"))
    save_file({"model.visual.merger.ln_q.weight": torch.zeros(3)}, str(d / "model-00002-of-00002.safetensors"))
    torch.save({"not": "weights"}, str(d / "training_args.bin"))
    torch.save({"master": torch.zeros(2)}, str(d / "optimizer.pt"))
    torch.save({"x": 1}, str(d / "rng_state.pth"))
    sd = read_checkpoint(str(d))
    assert set(sd) == {"model.norm.weight", "visual.merger.ln_q.weight"}
    save_file({"stale": torch.zeros(1)}, str(d / "model-old.safetensors"))
    (d / "model.safetensors.index.json").write_text(_json.dumps({"weight_map": {
        "a": "model-00001-of-00002.safetensors", "b": "model-00002-of-00002.safetensors"}}))
    assert set(read_checkpoint(str(d))) == {"model.norm.weight", "visual.merger.ln_q.weight"}
    with _pytest.raises(FileNotFoundError):
        read_checkpoint("Qwen/definitely-not-a-local-dir")
