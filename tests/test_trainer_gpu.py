"""GPU: SGRLVRTrainer end to end on the tiny model with a fake processor: two optimizer steps with T-GRPO on,
reward plugin API, metric keys, checkpoint round trip."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from fake_processor import FakeProcessor                                  # noqa: E402
from golden_util import load_tiny                                         # noqa: E402
from spacer_amd import kernels as K                                         # noqa: E402
from spacer_amd.open_r1.config import GRPOConfig, GRPOScriptArguments     # noqa: E402
from spacer_amd.open_r1.rewards import format_reward                      # noqa: E402
from spacer_amd.open_r1.trainer import SGRLVRTrainer                      # noqa: E402
from spacer_amd.qwen2vl.config import TINY                                # noqa: E402
from spacer_amd.qwen2vl.weights import FlatParams, load_state_dict        # noqa: E402

SEEN = []


def accuracy_reward(prompts, completions, video_path=None, **kw):
    """Plugin-API probe: checks what the trainer passes (TR:587-592) and returns content-dependent rewards."""
    SEEN.append(dict(n=len(completions), keys=sorted(kw), video_path=video_path is not None))
    assert len(prompts) == len(completions) == len(kw["problem_type"]) == len(kw["solution"])
    assert all(isinstance(c, list) and c[0]["role"] == "assistant" for c in completions)
    return [float(len(c[0]["content"]) % 3) for c in completions]


def test_trainer_two_steps(dev, tmp_path):
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    before = params.flat.clone()
    rows = []
    for i in range(2):
        frames = torch.randint(0, 256, (6, 3, 56, 84), generator=torch.Generator().manual_seed(i), dtype=torch.uint8)
        rows.append(dict(prompt=[{"role": "user", "content": [{"type": "video"}, {"type": "text", "text": f"what is in clip {i} ?"}]}],
                         path=frames, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>",
                         problem_id=i, options=["A. x", "B. y"], data_source="other"))
    args = GRPOConfig(output_dir=str(tmp_path), max_completion_length=8, num_generations=4, learning_rate=1e-4, max_steps=2,
                      logging_steps=1, save_steps=2, seed=3)
    trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                            script_args=GRPOScriptArguments(temporal=True, len_control=True), train_dataset=rows,
                            processing_class=FakeProcessor(TINY), device=dev)
    with pytest.raises(ValueError):
        trainer.compute_loss(None, [rows[0]], return_outputs=True)
    out = trainer.train()
    assert out["global_step"] == 2
    assert not torch.equal(before, trainer.engine.policy.flat)                     # weights moved
    assert torch.equal(trainer.engine.ref.flat, before)                            # reference model frozen
    # reward plugin saw K rollouts with video_path and K/2 shuffled rollouts without it (TR:571 vs :592)
    assert {(s["n"], s["video_path"]) for s in SEEN} == {(4, True), (2, False)}
    assert set(SEEN[0]["keys"]) >= {"path", "problem_type", "solution", "problem_id", "data_type"}
    logs = [json.loads(line) for line in open(os.path.join(str(tmp_path), "trainer_log.jsonl"))]
    assert len(logs) == 2
    for key in ("completion_length", "rewards/accuracy_reward", "rewards/format_reward", "all_wrong", "all_correct",
                "temporal_rewards", "reward", "reward_std", "kl", "loss", "learning_rate"):
        assert key in logs[0], key
    assert logs[0]["completion_length"] <= 8 and logs[0]["kl"] >= 0
    # checkpoint round trip in the original checkpoint names
    ck = os.path.join(str(tmp_path), "checkpoint-2")
    assert os.path.exists(os.path.join(ck, "model.safetensors"))
    t2 = SGRLVRTrainer(model=ck, reward_funcs=[format_reward], args=GRPOConfig(output_dir=str(tmp_path), max_steps=1),
                       train_dataset=rows, processing_class=FakeProcessor(TINY), device=dev, model_config=TINY)
    assert torch.equal(t2.engine.policy.flat, trainer.engine.policy.flat)


def test_accumulation_micro_batches_roll_out_together(dev, tmp_path):
    """All micro-batches of a gradient-accumulation step sample from the same weights: ONE generate call (prompts + T-GRPO
    twins of every micro-batch in one decode batch), then per-sample rewards / scoring / backward, one optimizer step."""
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    rows = []
    for i in range(4):
        frames = torch.randint(0, 256, (6, 3, 56, 84), generator=torch.Generator().manual_seed(10 + i), dtype=torch.uint8)
        rows.append(dict(prompt=[{"role": "user", "content": [{"type": "video"}, {"type": "text", "text": f"clip {i} ?"}]}],
                         path=frames, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>",
                         problem_id=i, options=["A. x", "B. y"], data_source="other"))
    args = GRPOConfig(output_dir=str(tmp_path), max_completion_length=6, num_generations=4, learning_rate=1e-4, max_steps=2,
                      gradient_accumulation_steps=2, logging_steps=1, save_steps=0, seed=5)
    trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                            script_args=GRPOScriptArguments(temporal=True, len_control=True), train_dataset=rows,
                            processing_class=FakeProcessor(TINY), device=dev)
    calls = []
    inner = trainer._generate
    trainer._generate = lambda prompts, n, sp: (calls.append(len(prompts)), inner(prompts, n, sp))[1]
    SEEN.clear()
    assert trainer.train()["global_step"] == 2
    assert calls == [4, 4]                                  # per optimizer step: 2 samples x (prompt + shuffled twin)
    assert sorted((s["n"], s["video_path"]) for s in SEEN) == [(2, False)] * 4 + [(4, True)] * 4
    logs = [json.loads(line) for line in open(os.path.join(str(tmp_path), "trainer_log.jsonl"))]
    assert len(logs) == 2 and all(lg["kl"] >= 0 for lg in logs)


def _video_rows(n, seed0=20):
    rows = []
    for i in range(n):
        frames = torch.randint(0, 256, (6, 3, 56, 84), generator=torch.Generator().manual_seed(seed0 + i), dtype=torch.uint8)
        rows.append(dict(prompt=[{"role": "user", "content": [{"type": "video"}, {"type": "text", "text": f"clip {i} ?"}]}],
                         path=frames, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>",
                         problem_id=i, options=["A. x", "B. y"], data_source="other"))
    return rows


def test_image_rows_train(dev, tmp_path):
    """``data_type == "image"`` rows (SG-RLVR.py:319-352, TR:396-414): the image goes through fetch_image (PIL, as the
    reference), the prompt carries <|image_pad|> placeholders + image_grid_thw, T-GRPO is skipped (temporal_rewards 0.5)."""
    from PIL import Image
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    rows = []
    for i in range(2):
        arr = torch.randint(0, 256, (60, 90, 3), generator=torch.Generator().manual_seed(40 + i), dtype=torch.uint8).numpy()
        rows.append(dict(prompt=[{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": f"what is in image {i} ?"}]}],
                         path=Image.fromarray(arr), data_type="image", problem_type="multiple choice", solution="<answer>A</answer>",
                         problem_id=i, options=["A. x", "B. y"], data_source="other"))
    args = GRPOConfig(output_dir=str(tmp_path), max_completion_length=6, num_generations=4, learning_rate=1e-4, max_steps=2,
                      logging_steps=1, save_steps=0, seed=3)
    trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                            script_args=GRPOScriptArguments(temporal=True, len_control=True), train_dataset=rows,
                            processing_class=FakeProcessor(TINY), device=dev)
    prep = trainer._prepare([rows[0]])
    assert prep["sproc"] is None and not prep["has_video"] and "pixel_values" in prep["proc"]
    pi = trainer._prompt_input(prep["proc"])
    assert int((pi.ids == TINY.image_token_id).sum()) == pi.grids[0][1] * pi.grids[0][2] // 4 and pi.grids[0][0] == 1
    SEEN.clear()
    assert trainer.train()["global_step"] == 2
    assert {(s["n"], s["video_path"]) for s in SEEN} == {(4, True)}              # no shuffled twin for images
    logs = [json.loads(line) for line in open(os.path.join(str(tmp_path), "trainer_log.jsonl"))]
    assert all(lg["temporal_rewards"] == 0.5 and lg["kl"] >= 0 and lg["loss"] == lg["loss"] for lg in logs)


def test_resume_skips_consumed_steps_and_restores_optimizer(dev, tmp_path, monkeypatch):
    """--resume_from_checkpoint (SG-RLVR.py:377-381, HF Trainer semantics): the run continues AFTER the samples already
    consumed (no replay from position 0), and with --save_only_model false the fp32 master weights and Adam moments come back,
    so a resumed run reproduces the uninterrupted one."""
    monkeypatch.setattr(K.PLAN, "skinny_blocks", 1)          # decode GEMMs without split-K atomics: reproducible rollouts
    g = load_tiny()
    rows = _video_rows(3)
    common = dict(reward_funcs=[accuracy_reward, format_reward], script_args=GRPOScriptArguments(temporal=False, len_control=True),
                  train_dataset=rows, processing_class=FakeProcessor(TINY), device=dev)

    def fresh():
        p = FlatParams.empty(TINY, dev)
        load_state_dict(p, g["w"])
        return p

    def cfg(out, steps):
        return GRPOConfig(output_dir=out, max_completion_length=6, num_generations=4, learning_rate=1e-3, max_steps=steps,
                          logging_steps=1, save_steps=2, save_only_model=False, seed=9, use_decode_graph=False)

    order = []
    full = SGRLVRTrainer(model=fresh(), args=cfg(str(tmp_path / "full"), 3), **common)
    inner = full._prepare
    full._prepare = lambda inputs, seed=0: (order.append(inputs[0]["problem_id"]), inner(inputs, seed))[1]
    full.train()
    ck = str(tmp_path / "full" / "checkpoint-2")
    assert os.path.exists(os.path.join(ck, "optimizer.pt"))
    seen = []
    res = SGRLVRTrainer(model=fresh(), args=cfg(str(tmp_path / "resumed"), 3), model_config=TINY, **common)
    inner2 = res._prepare
    res._prepare = lambda inputs, seed=0: (seen.append(inputs[0]["problem_id"]), inner2(inputs, seed))[1]
    assert res.train(resume_from_checkpoint=ck)["global_step"] == 3
    assert seen == order[2:3], (seen, order)                       # only the third sample of the epoch's permutation is trained
    assert res.engine.step_count == 3
    # same sample, same rollout seed, restored master + moments -> the same weights up to backward-atomics rounding
    a, b = full.engine.policy.flat.float(), res.engine.policy.flat.float()
    assert float((a - b).abs().max()) <= 2e-3 * float(a.abs().max())
    assert float((full.engine.m - res.engine.m).abs().max()) <= 1e-2 * float(full.engine.m.abs().max()) + 1e-8


def test_gpu_front_end_equals_the_processor_route(dev, tmp_path):
    """Video rows: frames sampled -> resized on the GPU (spacer_resize_bicubic_aa_u8) -> K1 patchify, against the reference's
    route (torch bicubic-antialias resize on the CPU + the processor's patchify): same token ids, same grid, and the same
    bf16 pixel rows except where the GPU resize rounds a .5 tie the other way (< 1e-3 of the values, one uint8 level)."""
    g = load_tiny()
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    frames = torch.randint(0, 256, (40, 3, 120, 160), generator=torch.Generator().manual_seed(77), dtype=torch.uint8)
    row = dict(prompt=[{"role": "user", "content": [{"type": "video"}, {"type": "text", "text": "what moves ?"}]}],
               path=frames, data_type="video", problem_type="multiple choice", solution="<answer>A</answer>", problem_id=0,
               options=["A. x", "B. y"], data_source="other")
    args = GRPOConfig(output_dir=str(tmp_path), max_completion_length=4, num_generations=2, max_steps=1, save_steps=0)
    common = dict(model=params, reward_funcs=[format_reward], args=args, script_args=GRPOScriptArguments(temporal=True),
                  train_dataset=[row], device=dev)
    nat = SGRLVRTrainer(processing_class=FakeProcessor(TINY), **common)
    ref = SGRLVRTrainer(processing_class=FakeProcessor(TINY, with_tokenizer=False), **common)
    pa, pb = nat._prepare([row], 5), ref._prepare([row], 5)
    assert pa["proc"].get("native") and not (hasattr(pb["proc"], "get") and pb["proc"].get("native"))
    for key in ("proc", "sproc"):
        a, b = nat._prompt_input(pa[key]), ref._prompt_input(pb[key])
        assert torch.equal(a.ids, b.ids) and list(a.grids) == list(b.grids)
        d = (a.pix.float() - b.pix.float()).abs()
        assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) < 0.08, (float((d > 0).float().mean()), float(d.max()))
    assert not torch.equal(nat._prompt_input(pa["proc"]).pix, nat._prompt_input(pa["sproc"]).pix)      # the twin is shuffled


def test_groups_per_pass_does_not_change_the_step(dev, tmp_path):
    """Three gradient-accumulation micro-batches scored one per pass, two per pass ([2, 1]) and three per pass: the same rollouts
    (same seeds), the same rewards and metrics, and -- compared BEFORE the optimizer (one Adam step is scale-invariant and would
    hide a wrong micro-batch weight) -- the same accumulated gradient and gradient norm up to the summation order of the
    backward's atomics: every micro-batch weighs 1 / gradient_accumulation_steps whatever the pass holds."""
    g = load_tiny()
    rows = _video_rows(3, seed0=70)
    finals, logs, grads = {}, {}, {}
    for gpp in (1, 2, 3):
        params = FlatParams.empty(TINY, dev)
        load_state_dict(params, g["w"])
        out = tmp_path / f"gpp{gpp}"
        args = GRPOConfig(output_dir=str(out), max_completion_length=6, num_generations=4, learning_rate=1e-4, max_steps=1,
                          gradient_accumulation_steps=3, logging_steps=1, save_steps=0, seed=11, groups_per_pass=gpp)
        trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                                script_args=GRPOScriptArguments(temporal=True, len_control=True), train_dataset=rows,
                                processing_class=FakeProcessor(TINY), device=dev)
        inner = trainer.engine.optimizer_step

        def spy(world=1, inner=inner, eng=trainer.engine, gpp=gpp):
            grads[gpp] = eng.G.flat.clone()
            return inner(world)
        trainer.engine.optimizer_step = spy
        assert trainer.train()["global_step"] == 1
        finals[gpp] = trainer.engine.master.flat.clone()
        logs[gpp] = json.loads(open(os.path.join(str(out), "trainer_log.jsonl")).readline())
    assert float(grads[1].norm()) > 0
    for gpp in (2, 3):
        for key in ("completion_length", "rewards/accuracy_reward", "rewards/format_reward", "reward", "reward_std", "temporal_rewards"):
            assert logs[gpp][key] == logs[1][key], (gpp, key)
        assert abs(logs[gpp]["loss"] - logs[1]["loss"]) < 1e-5 and abs(logs[gpp]["kl"] - logs[1]["kl"]) < 1e-6
        rel = float((grads[gpp] - grads[1]).norm() / grads[1].norm())
        assert rel < 2e-3, (gpp, rel)                                  # a 2x / (2/3, 2/3, 1/3) weighting gives 1.0 / 0.3 here
        assert abs(logs[gpp]["grad_norm"] / logs[1]["grad_norm"] - 1.0) < 2e-3, (gpp, logs[gpp]["grad_norm"], logs[1]["grad_norm"])
        d = float((finals[gpp] - finals[1]).abs().max())
        assert d <= 2.1e-4, d          # one AdamW step at lr 1e-4 moves a weight by <= 1e-4: sign flips of ~0 gradients bound the gap


def test_mixed_text_and_vision_rows_fall_back_to_separate_passes(dev, tmp_path):
    """with groups_per_pass = 2 a text-only row next to a video row must not land in one token-packed pass (a pass
    runs one ViT over its groups): the trainer splits such neighbours and still takes the step."""
    g = load_tiny()
    rows = _video_rows(2, seed0=90)
    text_row = dict(rows[1], data_type="text", path="", prompt=[{"role": "user", "content": [{"type": "text", "text": "which one ?"}]}])
    params = FlatParams.empty(TINY, dev)
    load_state_dict(params, g["w"])
    args = GRPOConfig(output_dir=str(tmp_path), max_completion_length=6, num_generations=4, learning_rate=1e-4, max_steps=1,
                      gradient_accumulation_steps=2, logging_steps=1, save_steps=0, seed=5, groups_per_pass=2)
    trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                            script_args=GRPOScriptArguments(temporal=False, len_control=True), train_dataset=[rows[0], text_row],
                            processing_class=FakeProcessor(TINY), device=dev)
    assert trainer.train()["global_step"] == 1
    assert any("separate passes" in line for line in trainer._log_lines)
    lg = json.loads(open(os.path.join(str(tmp_path), "trainer_log.jsonl")).readline())
    assert lg["loss"] == lg["loss"] and lg["grad_norm"] > 0


def test_precise_logps_flag_trains_through_the_trainer(dev, tmp_path, monkeypatch):
    """--precise_logps true (GRPOConfig.precise_logps -> GRPOHyper.precise_logps): the trainer's step takes its log-probs, KL and loss
    from the precise mode and still moves the weights; the logged loss / KL of step 1 differ from the fast trainer's by less than the
    fast path's own log-prob floor, and the injected-engine check refuses a mismatching engine."""
    monkeypatch.setattr(K.PLAN, "skinny_blocks", 1)          # decode GEMMs without split-K atomics: the two runs sample the same rollouts
    g = load_tiny()
    rows = _video_rows(2, seed0=40)
    logs = {}
    for mode in (False, True):
        params = FlatParams.empty(TINY, dev)
        load_state_dict(params, g["w"])
        before = params.flat.clone()
        out = tmp_path / ("precise" if mode else "fast")
        args = GRPOConfig(output_dir=str(out), max_completion_length=6, num_generations=4, learning_rate=1e-4, max_steps=2,
                          gradient_accumulation_steps=2, logging_steps=1, save_steps=0, seed=21, precise_logps=mode)
        trainer = SGRLVRTrainer(model=params, reward_funcs=[accuracy_reward, format_reward], args=args,
                                script_args=GRPOScriptArguments(temporal=False, len_control=True), train_dataset=rows,
                                processing_class=FakeProcessor(TINY), device=dev)
        assert trainer.engine.h.precise_logps is mode
        assert trainer.train()["global_step"] == 2
        assert not torch.equal(before, trainer.engine.policy.flat)
        logs[mode] = [json.loads(line) for line in open(os.path.join(str(out), "trainer_log.jsonl"))]
        if mode:
            assert any("precise_logps" in line for line in trainer._log_lines)
            with pytest.raises(ValueError):
                SGRLVRTrainer(model=params, reward_funcs=[format_reward], args=GRPOConfig(output_dir=str(out), num_generations=4),
                              train_dataset=rows, processing_class=FakeProcessor(TINY), device=dev, engine=trainer.engine)
    a, b = logs[False][0], logs[True][0]
    assert a["completion_length"] == b["completion_length"] and a["reward"] == b["reward"]          # same rollouts (same seeds, step 1)
    assert abs(a["loss"] - b["loss"]) < 5e-3 and abs(a["kl"] - b["kl"]) < 5e-3 and b["grad_norm"] > 0
