"""``SGRLVRTrainer``: the drop-in surface of the reference trainer on the MI355X-native engine.

Same class name, constructor keywords, ``train`` / ``compute_loss`` / ``save_model`` / ``log`` methods and metric keys
as /root/reference/SpaceR-SG-RLVR/src/r1-v/src/open_r1/trainer/SG_RLVR_trainer.py (TR: ctor :137-153, compute_loss
:384-686, log :688-695).  What differs, by design:
  * no HF Trainer / DeepSpeed / trl underneath: the loop, AdamW and the data-parallel exchange are this repo's
    (spacer_amd/grpo.py); ZeRO-3 is replaced by full replicas (288 GB HBM) + one flat RCCL all-reduce;
  * ``compute_loss`` also back-propagates (rollout -> scoring -> loss -> backward are fused into one pass over the
    group's tape); it returns the loss tensor for logging;
  * the nine ``gather_for_metrics`` calls (TR:650-683) are one packed all-gather;
  * the silent try/except fallbacks (TR:405-414, 526-547) are NOT reproduced: errors surface.
``inputs`` = list of dataset rows with keys prompt, path, data_type, problem_type, solution, problem_id, ...
Reward functions keep the reference plugin signature f(prompts=, completions=, video_path=, **columns) -> list[float].
"""
from __future__ import annotations

import copy
import json
import os
import time
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Dict, List, Optional, Sequence, Union

import torch

from ...grpo import GRPOEngine, GRPOHyper, group_advantages, length_bonus, temporal_bonus
from ...qwen2vl.checkpoint import config_of_dir, read_checkpoint, write_checkpoint
from ...qwen2vl.config import Qwen2VLConfig, preset_for
from ...qwen2vl.weights import FlatParams, load_state_dict
from ...rollout import PromptInput, SamplingParams
from ..config import GRPOConfig, GRPOScriptArguments

RewardFunc = Callable[..., List[float]]


# ----------------------------------------------------------------------------------------- prompt text
def is_conversational(example: dict) -> bool:
    p = example.get("prompt")
    return isinstance(p, list) and len(p) > 0 and isinstance(p[0], dict) and "role" in p[0]


def qwen2vl_chat_template(messages: Sequence[dict], add_generation_prompt: bool = True) -> str:
    """Qwen2-VL's ChatML template (what ``processing_class.apply_chat_template`` renders): a default system turn,
    ``<|vision_start|><|video_pad|><|vision_end|>`` / image markers for media parts, text parts verbatim."""
    out = []
    if not messages or messages[0].get("role") != "system":
        out.append("<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n")
    for m in messages:
        out.append(f"<|im_start|>{m['role']}\n")
        c = m.get("content")
        if isinstance(c, str):
            out.append(c)
        else:
            for part in c or []:
                t = part.get("type")
                if t == "video" or "video" in part:
                    out.append("<|vision_start|><|video_pad|><|vision_end|>")
                elif t == "image" or "image" in part:
                    out.append("<|vision_start|><|image_pad|><|vision_end|>")
                elif "text" in part and part["text"] is not None:
                    out.append(part["text"])
        out.append("<|im_end|>\n")
    if add_generation_prompt:
        out.append("<|im_start|>assistant\n")
    return "".join(out)


def maybe_apply_chat_template(example: dict, processing_class) -> Dict[str, str]:
    if not is_conversational(example):
        return {"prompt": example["prompt"]}
    if hasattr(processing_class, "apply_chat_template"):
        return {"prompt": processing_class.apply_chat_template(example["prompt"], tokenize=False, add_generation_prompt=True)}
    return {"prompt": qwen2vl_chat_template(example["prompt"], add_generation_prompt=True)}


def remove_none_from_data(data):
    """TR:368-376: datasets.map fills absent keys of content parts with None; drop them."""
    for entry in data:
        if isinstance(entry.get("content"), list):
            for sub in entry["content"]:
                if isinstance(sub, dict):
                    for k in [k for k, v in sub.items() if v is None]:
                        del sub[k]
    return data


def repeat_columns(inputs: List[dict], n: int) -> Dict[str, list]:
    """TR:587-591: every dataset column except prompt/completion, each value repeated n times."""
    out: Dict[str, list] = {k: [] for k in inputs[0].keys() if k not in ("prompt", "completion")}
    for k in out:
        for ex in inputs:
            out[k].extend([ex[k]] * n)
    return out


# ----------------------------------------------------------------------------------------- packed metrics (C2)
METRIC_SLOTS = 16


def pack_metrics(completion_lengths, rewards_per_func, rewards, temporal_reward, std, mean_kl) -> torch.Tensor:
    """One fp32 vector per rank instead of nine gathers: [n_funcs, mean len, mean reward, mean std, kl, temporal,
    all_wrong flag, all_correct flag, per-func means ...]."""
    nf = rewards_per_func.shape[1]
    assert 8 + nf <= METRIC_SLOTS
    v = torch.zeros(METRIC_SLOTS, dtype=torch.float32)
    v[0] = nf
    v[1] = float(completion_lengths.float().mean())
    v[2] = float(rewards.mean())
    v[3] = float(std.mean())
    v[4] = float(mean_kl)
    v[5] = float(temporal_reward)
    v[6] = float(bool((rewards <= 1).all()))          # TR:665
    v[7] = float(bool((rewards >= 2).all()))          # TR:668
    v[8:8 + nf] = rewards_per_func.float().mean(0)
    return v


def reduce_metrics(stacked: torch.Tensor, func_names: Sequence[str], temporal: bool) -> Dict[str, float]:
    """stacked [world, METRIC_SLOTS] -> the reference's metric dict (TR:650-683): means over ranks; all_wrong /
    all_correct are the FRACTION OF RANKS whose whole group scored <= 1 / >= 2."""
    m = stacked.float().mean(0)
    out = {"completion_length": float(m[1])}
    for i, n in enumerate(func_names):
        out[f"rewards/{n}"] = float(m[8 + i])
    # (TR:665-669: integer counts of ranks divided in Python doubles -- 1 / 3 is 0.3333333333333333 in the reference's log, not the fp32 mean
    # 0.3333333432674408; pinned by tests/golden/grpo_lines.json "metrics")
    world = stacked.shape[0]
    out["all_wrong"] = int(stacked[:, 6].sum().item()) / world
    out["all_correct"] = int(stacked[:, 7].sum().item()) / world
    if temporal:
        out["temporal_rewards"] = float(m[5])
    out["reward"] = float(m[2])
    out["reward_std"] = float(m[3])
    out["kl"] = float(m[4])
    return out


def shard_indices(n_rows: int, rank: int, world: int, seed: int, epoch: int, shuffle: bool = True) -> List[int]:
    """DistributedSampler semantics: one shared permutation per epoch, padded to a multiple of world, rank-strided."""
    g = torch.Generator().manual_seed(seed + epoch)
    order = torch.randperm(n_rows, generator=g).tolist() if shuffle else list(range(n_rows))
    total = (n_rows + world - 1) // world * world
    order += order[:total - n_rows]
    return order[rank:total:world]


# ----------------------------------------------------------------------------------------- the trainer
class SGRLVRTrainer:
    def __init__(self, model: Union[str, FlatParams], reward_funcs: Union[RewardFunc, List[RewardFunc]],
                 args: Optional[GRPOConfig] = None, script_args: Optional[GRPOScriptArguments] = None, train_dataset=None,
                 eval_dataset=None, processing_class=None, reward_processing_classes=None, callbacks=None,
                 optimizers=(None, None), peft_config=None, max_pixels: Optional[int] = 12845056,
                 min_pixels: Optional[int] = 3136, attn_implementation: str = "flash_attention_2", *,
                 model_config: Optional[Qwen2VLConfig] = None, device=None, process_group=None, engine: Optional[GRPOEngine] = None):
        if args is None:
            name = model if isinstance(model, str) else "model"
            args = GRPOConfig(output_dir=f"{name.split('/')[-1]}-GRPO")
        self.args = args
        self.script_args = script_args or GRPOScriptArguments()
        if peft_config is not None:
            raise NotImplementedError("PEFT adapters are outside the SG-RLVR hot path of this engine")
        if any(isinstance(f, str) for f in (reward_funcs if isinstance(reward_funcs, list) else [reward_funcs])):
            # The reference still LOADS a string as AutoModelForSequenceClassification (TR:240-244), but its compute_loss calls every
            # reward function as ``reward_func(prompts=, completions=, video_path=, **columns)`` (TR:579-592): TRL's
            # ``isinstance(reward_func, PreTrainedModel)`` branch (tokenise prompt + completion, read logits[:, 0]) was removed there, so a
            # reward MODEL fails at the reference's first step with a TypeError.  Refused here at construction, with the reason.
            raise NotImplementedError("string reward_funcs (sequence-classification reward models): the reference loads them (TR:240-244) but its "
                                      "compute_loss calls every reward function as a plain callable (TR:579-592), which a PreTrainedModel is not; "
                                      "pass Python callables (open_r1.rewards.accuracy_reward / format_reward, or your own)")
        if attn_implementation not in (None, "flash_attention_2", "sdpa", "eager"):
            raise ValueError(f"unknown attn_implementation {attn_implementation!r}")
        self._log_lines: List[str] = []
        self._note(f"attn_implementation={attn_implementation!r} accepted and ignored: attention is libspacer_hip's flash kernel")
        if getattr(args, "deepspeed", None):
            self._note(f"--deepspeed {args.deepspeed} accepted and ignored: full replicas + flat RCCL all-reduce replace ZeRO-3")
        self.device = device or torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        self.pg = process_group
        self.world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
        self.rank = torch.distributed.get_rank(process_group) if process_group is not None else 0

        # ---- model (policy) and reference model (a frozen copy, TR:205-217)
        if isinstance(model, str):
            # architecture: the snapshot's own config.json when `model` is a directory, else the preset its name selects
            cfg = model_config or config_of_dir(model) or preset_for(model)
            params = FlatParams.empty(cfg, self.device)
            load_state_dict(params, read_checkpoint(model))
            self.model_id = model
        else:
            params, cfg = model, (model_config or model.cfg)
            self.model_id = getattr(model, "name", "in-memory")
        self.cfg = cfg
        self.processing_class = processing_class
        if processing_class is None:
            raise ValueError("processing_class is required (AutoProcessor of the checkpoint, or any object with the same "
                             "__call__ / batch_decode / eos_token_id / pad_token_id surface)")
        self.reward_funcs = reward_funcs if isinstance(reward_funcs, list) else [reward_funcs]
        self.reward_processing_classes = reward_processing_classes or [None] * len(self.reward_funcs)
        self.max_prompt_length = args.max_prompt_length
        self.max_completion_length = args.max_completion_length
        self.num_generations = args.num_generations
        self.shuffled_num_generations = self.num_generations // 2
        self.temporal = bool(self.script_args.temporal)
        if self.temporal and self.shuffled_num_generations < 1:
            # (TR:473 would ask HF generate for num_return_sequences = 0 on the shuffled twin)
            raise ValueError("--temporal true needs --num_generations >= 2: the frame-shuffled twin generates num_generations // 2 rollouts")
        self.len_control = bool(self.script_args.len_control)
        self.beta = args.beta
        self.era_rule = bool(getattr(self.script_args, "mrope_era_rule", True))
        # video rows: resize + patchify on the GPU (libspacer_hip) when the processor exposes its tokenizer; SPACER_FRONTEND=hf
        # keeps the reference's CPU route (torch resize + HF processor patchify) for A/B runs
        self.native_frontend = os.environ.get("SPACER_FRONTEND", "native") != "hf"
        self.train_dataset, self.eval_dataset = train_dataset, eval_dataset
        n_rows = len(train_dataset) if train_dataset is not None else 0
        per_step = max(1, self.world * args.per_device_train_batch_size * args.gradient_accumulation_steps)
        total_steps = args.max_steps if args.max_steps > 0 else max(1, int(n_rows * args.num_train_epochs) // per_step)
        hyper = GRPOHyper(num_generations=args.num_generations, beta=args.beta, learning_rate=args.learning_rate,
                          weight_decay=args.weight_decay, adam_beta1=args.adam_beta1, adam_beta2=args.adam_beta2,
                          adam_eps=args.adam_epsilon, max_grad_norm=args.max_grad_norm, temporal=self.temporal,
                          len_control=self.len_control, lr_scheduler_type=args.lr_scheduler_type, total_steps=total_steps,
                          warmup_steps=args.warmup_steps, recompute=bool(getattr(args, "gradient_checkpointing", False)),
                          grad_algo=getattr(args, "grad_algo", "allreduce"), precise_logps=bool(getattr(args, "precise_logps", False)),
                          reuse_prefill={"auto": None, "true": True, "false": False}.get(str(getattr(args, "reuse_prefill", "auto")).lower()))
        if hyper.precise_logps:
            self._note("--precise_logps true: policy / reference log-probs in the precise mode (hi+lo bf16 operand pairs; <= 1e-3 of fp32 "
                       "at full depth), gradient on the production backward of the same tape")
        if hyper.recompute:
            self._note("--gradient_checkpointing true: selective activation recompute (MLP intermediates + lm_head logits are "
                       "recomputed in the backward; gradients are bit-identical to the stored path)")
        self.total_steps = total_steps
        # ``engine``: an existing GRPOEngine over ``model`` (bench.py times the trainer on the engine it already holds: a second
        # copy of master weights + Adam state would not fit)
        if engine is not None:
            # an injected engine keeps ITS hyper-parameters: refuse a silent mismatch with what ``args`` asks for
            for key in ("num_generations", "beta", "max_grad_norm", "precise_logps"):
                if getattr(engine.h, key) != getattr(hyper, key):
                    raise ValueError(f"engine= was built with {key}={getattr(engine.h, key)!r} but args ask for {getattr(hyper, key)!r}")
        self.engine = engine if engine is not None else GRPOEngine(cfg, params, hyper, process_group=process_group)
        # micro-batches of a gradient-accumulation step scored / back-propagated per token-packed pass (GRPOEngine.
        # score_and_backward_multi): 2 = what 288 GB holds at 7B with 16-frame prompts and 512-token rollouts
        self.groups_per_pass = max(1, int(getattr(args, "groups_per_pass", 2)))
        self.suppress_eos = False            # bench.py's throughput mode: fixed-length rollouts
        self._metrics: Dict[str, list] = defaultdict(list)
        self.global_step = 0
        self._sample_seed = args.seed * 1000003 + self.rank

    # ------------------------------------------------------------------ small helpers
    def _note(self, msg: str) -> None:
        self._log_lines.append(msg)
        if int(os.environ.get("RANK", "0")) == 0:
            print("[spacer_amd] " + msg, flush=True)

    def _prompt_input(self, proc_out: dict) -> PromptInput:
        if proc_out.get("native"):
            return self._prompt_input_native(proc_out)
        ids = proc_out["input_ids"]
        if self.max_prompt_length is not None:                 # TR:432-440 (left truncation of ids)
            ids = ids[:, -self.max_prompt_length:]
        assert ids.shape[0] == 1, "one prompt per compute_loss call, as in the reference (per_device_train_batch_size 1)"
        ids = ids[0].to(self.device).long()
        pix, grids = None, None
        key = "pixel_values_videos" if "pixel_values_videos" in proc_out else ("pixel_values" if "pixel_values" in proc_out else None)
        if key is not None:
            pv = proc_out[key].to(self.device)
            pix = torch.zeros(pv.shape[0], self.cfg.patch_kpad, device=self.device, dtype=torch.bfloat16)
            pix[:, :pv.shape[1]] = pv.to(torch.bfloat16)
            g = proc_out["video_grid_thw" if key == "pixel_values_videos" else "image_grid_thw"]
            grids = [tuple(int(v) for v in row) for row in g.tolist()]
        sec = proc_out.get("second_per_grid_ts") if hasattr(proc_out, "get") else None
        if sec is not None:
            sec = [float(v) for v in (sec.tolist() if hasattr(sec, "tolist") else sec)]
        return PromptInput(ids=ids, pix=pix, grids=grids, second_per_grid_ts=sec)

    def _prompt_input_native(self, nat: dict) -> PromptInput:
        """GPU front end: sampled uint8 frames -> (H2D) -> bicubic-antialias resize -> rescale / normalise / patchify, all in
        libspacer_hip (spacer_resize_bicubic_aa_u8 + spacer_patchify = K1) instead of torchvision + the HF processor on the CPU
        (QU:310-315, TR:417-425).  ``perm``: the T-GRPO twin's temporal shuffle (TR:442-458), applied to the resized frames."""
        from ... import kernels as K
        from ...qwen_vl_utils.vision_process import resize_frames_gpu
        cfg = self.cfg
        frames = nat["frames_u8"].to(self.device, non_blocking=True)
        if tuple(frames.shape[2:]) != tuple(nat["hw"]):
            frames = resize_frames_gpu(frames, nat["hw"])
        if nat.get("perm") is not None:
            frames = frames[nat["perm"].to(self.device)].contiguous()
        pix, grid = K.patchify(frames, cfg.patch, cfg.tpatch, cfg.merge, cfg.patch_kpad)
        assert tuple(grid) == tuple(nat["grid"]), (grid, nat["grid"])
        ids = nat["input_ids"]
        if self.max_prompt_length is not None:
            ids = ids[:, -self.max_prompt_length:]
        return PromptInput(ids=ids[0].to(self.device).long(), pix=pix, grids=[tuple(grid)], second_per_grid_ts=nat.get("second_per_grid_ts"))

    def _prepare_native(self, inputs, prompts_text, conv, shuffle_seed: int) -> Optional[dict]:
        """Host half of the GPU front end for a video row: frame sampling plan + tokenisation with the placeholder expanded
        to the number of merged video tokens (what the HF processor does before it tokenises).  None -> use the processor."""
        from ...qwen_vl_utils.vision_process import sample_video
        tok = getattr(self.processing_class, "tokenizer", None)
        part = conv[0]["content"][0]
        if tok is None or not self.native_frontend or len(prompts_text) != 1 or prompts_text[0].count("<|video_pad|>") != 1:
            return None
        got = sample_video(part)
        if got is None:
            return None
        frames, hw, _ = got
        cfg = self.cfg
        gt = (frames.shape[0] + cfg.tpatch - 1) // cfg.tpatch
        grid = (gt, hw[0] // cfg.patch, hw[1] // cfg.patch)
        nv = grid[0] * grid[1] * grid[2] // (cfg.merge ** 2)
        text = prompts_text[0].replace("<|video_pad|>", "<|video_pad|>" * nv)
        ids = tok([text], add_special_tokens=False)["input_ids"]
        ids = torch.as_tensor(ids, dtype=torch.long).view(1, -1)
        assert int((ids == cfg.video_token_id).sum()) == nv, "tokenizer did not keep one id per <|video_pad|>"
        sec = [cfg.tpatch / 2.0] if cfg.vit_kind == "qwen2_5" else None     # HF video processor default fps 2.0, no fps passed (TR:417-425)
        main = dict(native=True, input_ids=ids, frames_u8=frames, hw=hw, grid=grid, second_per_grid_ts=sec)
        twin = None
        if self.temporal:
            perm = torch.randperm(frames.shape[0], generator=torch.Generator().manual_seed(shuffle_seed))
            twin = dict(main, perm=perm)
        return dict(proc=main, sproc=twin, has_video=True)

    def _run_rewards(self, inputs, prompts, completion_ids, n, video_path=None) -> torch.Tensor:
        """TR:576-593 (and the shuffled twin :554-572): decode, wrap, call every reward function."""
        texts = self.processing_class.batch_decode(completion_ids.cpu(), skip_special_tokens=True)
        completions = [[{"role": "assistant", "content": t}] for t in texts] if is_conversational(inputs[0]) else texts
        rep_prompts = [p for p in prompts for _ in range(n)]
        out = torch.zeros(len(rep_prompts), len(self.reward_funcs), dtype=torch.float32)
        for i, fn in enumerate(self.reward_funcs):
            kw = repeat_columns(inputs, n)
            if video_path is not None:
                vals = fn(prompts=rep_prompts, completions=completions, video_path=video_path, **kw)
            else:
                vals = fn(prompts=rep_prompts, completions=completions, **kw)
            out[:, i] = torch.tensor([float(v) for v in vals], dtype=torch.float32)
        return out

    def _generate(self, prompts: List[PromptInput], n, sp: SamplingParams) -> torch.Tensor:
        """TR:463-481: n sampled completions per prompt (an int, or one count per prompt: the shuffled twins take G // 2, TR:473),
        int64 [sum of counts, C].  The rollout-server trainer overrides this."""
        # (a prefill tape that does not fit whole is kept for the first scoring passes' prompts, in whole passes: RolloutEngine._tape_keep_count)
        self.engine.roll.prefill_pass_size = self.groups_per_pass
        return self.engine.roll.generate(prompts, n, sp, use_graph=self.args.use_decode_graph)

    # ------------------------------------------------------------------ the step (TR:384-686)
    def _prepare(self, inputs, shuffle_seed: int = 0) -> dict:
        """Host half of a step (TR:395-430, and the shuffled twin's inputs TR:442-460): chat template, frame decode + resize
        (qwen_vl_utils), tokenisation + HF patchify.  CPU only, no engine state -- ``train`` runs it for the NEXT sample on a
        worker thread while the GPU works on the current one, so the step no longer starts with a CPU stall."""
        from ...qwen_vl_utils.vision_process import process_vision_info
        prompts_text = [maybe_apply_chat_template(ex, self.processing_class)["prompt"] for ex in inputs]
        conv = remove_none_from_data(copy.deepcopy(inputs[0]["prompt"]))
        if inputs[0]["data_type"] in ("image", "video"):
            conv[0]["content"][0][inputs[0]["data_type"]] = inputs[0]["path"]
        if inputs[0]["data_type"] == "video":
            nat = self._prepare_native(inputs, prompts_text, conv, shuffle_seed)
            if nat is not None:
                return nat
        image_inputs, video_inputs, _ = process_vision_info(conv, return_video_kwargs=True)
        call = dict(return_tensors="pt", padding=True, padding_side="left", add_special_tokens=False)
        proc = self.processing_class(text=copy.deepcopy(prompts_text), images=image_inputs, videos=video_inputs, **call)
        sproc = None
        if self.temporal and video_inputs:                                                   # T-GRPO twin (TR:442-460)
            perm = torch.randperm(video_inputs[0].size(0), generator=torch.Generator().manual_seed(shuffle_seed))
            sproc = self.processing_class(text=copy.deepcopy(prompts_text), images=image_inputs,
                                          videos=[video_inputs[0][perm]], **call)
        return dict(proc=proc, sproc=sproc, has_video=bool(video_inputs))

    def _rollout(self, preps: List[dict]) -> List[dict]:
        """TR:463-481 for one or SEVERAL prepared samples in one generate call.  The reference calls generate per sample (and a
        second time with G/2 for the frame-shuffled twin).  Decoding is bound by streaming the weights, not by the number of
        rows, so every prompt handed in -- the twins too, and all micro-batches of a gradient-accumulation step, which sample
        from the same weights -- decodes as ONE batch; a twin's surplus G/2 rollouts are dropped, so it costs its prefill,
        not a second decode loop.  Returns per sample dict(prompt, completion_ids [G, C], shuffled_ids [G/2, C] or None)."""
        G = self.num_generations
        sp = SamplingParams(max_new_tokens=self.max_completion_length, top_k=self.args.top_k, top_p=0.95, temperature=1.0,
                            seed=self._sample_seed + 7919 * self.global_step, era_rule=self.era_rule, suppress_eos=self.suppress_eos)
        # batch order: every sample's own prompt first, the frame-shuffled twins behind them -- the scored prompts are then CONSECUTIVE
        # in the prefill pass, which is what lets the policy's scoring pass take its prompt rows from a kept prefill tape
        # (Qwen2VLEngine._prefill_usable; interleaved [p0, twin0, p1, twin1] never qualified with two groups per pass: ADVICE r5)
        prompts = [self._prompt_input(prep["proc"]) for prep in preps]
        twin_at = {}
        for i, prep in enumerate(preps):
            if prep["sproc"] is not None:
                twin_at[i] = len(prompts)
                prompts.append(self._prompt_input(prep["sproc"]))
        # the twins generate G // 2 rollouts each, as the reference (TR:473): 8 x 8 + 8 x 4 = 96 decode rows per 8 samples instead of 128
        # (a token-step costs 4.87 instead of 5.47 ms; until round 6 every prompt decoded G rows and the twins' surplus was dropped)
        sg, n_main = self.shuffled_num_generations, len(preps)
        counts = [G] * n_main + [sg] * (len(prompts) - n_main)
        self.engine.roll.prefill_scored = n_main          # (the twins behind them are never scored: no tape kept for them)
        ids = self._generate(prompts, counts if len(prompts) > n_main else G, sp)
        host = ids.cpu()       # ONE device-to-host copy for the step, taken when the decode loop has just ended: the reward
        #                        functions read it while the scoring passes run (no sync inside the scoring phase)
        out = []
        for i in range(len(preps)):
            twin = i in twin_at
            t0 = n_main * G + (twin_at[i] - n_main) * sg if twin else 0
            sl = slice(t0, t0 + sg) if twin else None
            out.append(dict(prompt=prompts[i], completion_ids=ids[i * G:(i + 1) * G], completion_host=host[i * G:(i + 1) * G],
                            shuffled_ids=ids[sl] if twin else None, shuffled_host=host[sl] if twin else None))
        return out

    def _shape(self, inputs, prep: dict, rolled: dict) -> dict:
        """Host half of TR:554-638 for one sample: decode -> reward functions (and the shuffled twin's) -> T-GRPO / length bonus
        -> group advantages.  Reads only the host copy of the ids."""
        G = self.num_generations
        prompts = [x["prompt"] for x in inputs]
        comp = rolled.get("completion_host")
        comp = comp if comp is not None else rolled["completion_ids"].cpu()
        shuf = rolled.get("shuffled_host")
        if shuf is None and rolled.get("shuffled_ids") is not None:
            shuf = rolled["shuffled_ids"].cpu()
        shuffled_rpf = None
        if shuf is not None:                                                                 # T-GRPO (TR:442-481, :554-572)
            shuffled_rpf = self._run_rewards(inputs, prompts, shuf, self.shuffled_num_generations)
        rewards_per_func = self._run_rewards(inputs, prompts, comp, G, video_path=inputs[0]["path"])
        rewards, temporal_reward = temporal_bonus(rewards_per_func, shuffled_rpf, self.temporal, prep["has_video"])
        eos = getattr(self.processing_class, "eos_token_id", self.cfg.eos_token_id)
        is_eos = comp == eos
        lengths = torch.where(is_eos.any(1), is_eos.int().argmax(1) + 1, torch.full((G,), comp.shape[1]))
        rewards = length_bonus(rewards, rewards_per_func, lengths, self.len_control)
        adv, std = group_advantages(rewards, G)
        return dict(adv=adv, std=std, lengths=lengths, rewards_per_func=rewards_per_func, rewards=rewards,
                    temporal_reward=temporal_reward)

    def _record_metrics(self, shaped: List[dict], kls: List[torch.Tensor]) -> None:
        """TR:650-683 for every micro-batch of the step at once: ONE device sync for the KL values and ONE packed all-gather
        (the reference: nine gathers per micro-batch); the logged means are the same as recording them one by one."""
        kl_host = torch.stack([k.reshape(()) for k in kls]).float().cpu()
        packed = torch.stack([pack_metrics(sh["lengths"], sh["rewards_per_func"], sh["rewards"], sh["temporal_reward"], sh["std"],
                                           float(kl_host[i])) for i, sh in enumerate(shaped)])                   # [n, SLOTS]
        if self.pg is not None:
            src = packed.to(self.device) if torch.distributed.get_backend(self.pg) == "nccl" else packed
            buf = [torch.zeros_like(src) for _ in range(self.world)]
            torch.distributed.all_gather(buf, src, group=self.pg)
            stacked = torch.stack(buf).cpu()                                                                   # [world, n, SLOTS]
        else:
            stacked = packed.unsqueeze(0)
        names = [getattr(f, "__name__", str(f)) for f in self.reward_funcs]
        for j in range(stacked.shape[1]):
            for k, v in reduce_metrics(stacked[:, j], names, self.temporal).items():
                self._metrics[k].append(v)

    def compute_loss(self, model, inputs, return_outputs=False, num_items_in_batch=None, *, grad_scale: float = 1.0,
                     prepared: Optional[dict] = None, rolled: Optional[dict] = None, last_micro_batch: bool = False):
        if return_outputs:
            raise ValueError("The GRPOTrainer does not support returning outputs")
        eng = self.engine
        prep = prepared if prepared is not None else self._prepare(inputs, self._sample_seed + 104729 * self.global_step)
        if rolled is None:                      # stand-alone call: this sample's rollouts only
            rolled = self._rollout([prep])[0]
        shaped = []

        def advantages():                       # runs after both forward passes are queued: rewards overlap the scoring
            shaped.append(self._shape(inputs, prep, rolled))
            return [shaped[0]["adv"]]
        # last_micro_batch: finished layer ranges of this backward go to the data-parallel reducer while it still runs
        res = eng.score_and_backward(rolled["prompt"], rolled["completion_ids"], advantages, grad_scale=grad_scale,
                                     era_rule=self.era_rule, last_group=last_micro_batch)
        self._record_metrics(shaped, [res["kl"]])
        return res["loss"]

    def _accumulate(self, rows: List[dict], preps: List[dict], rolled: List[dict]) -> float:
        """The gradient-accumulation micro-batches of ONE optimizer step (TR:384-686 called ``acc`` times by HF Trainer), on the
        path bench.py measures: ``groups_per_pass`` micro-batches per token-packed scoring pass, reward functions of a pass
        computed on the host while its forwards run, NO device sync until the step's backward passes are all queued (the KL /
        loss read-back and the metric all-gather happen once per optimizer step).  Returns the mean loss."""
        eng, acc = self.engine, len(rows)
        gpp = max(1, min(self.groups_per_pass, acc))
        shaped: List[Optional[dict]] = [None] * acc
        kls: List[Optional[torch.Tensor]] = [None] * acc
        losses = []
        # passes: consecutive micro-batches, at most ``gpp`` each, that either ALL carry vision inputs or none does (a token-packed
        # pass runs one ViT over its groups, Qwen2VLEngine.score_groups); a text-only row next to a video row falls back to a
        # pass of its own
        passes: List[List[int]] = []
        for j in range(acc):
            vis = rolled[j]["prompt"].pix is not None
            if passes and len(passes[-1]) < gpp and (rolled[passes[-1][0]]["prompt"].pix is not None) == vis:
                passes[-1].append(j)
            else:
                passes.append([j])
        if gpp > 1 and len(passes) > (acc + gpp - 1) // gpp and not getattr(self, "_warned_mixed_pass", False):
            self._warned_mixed_pass = True
            self._note("text-only and vision rows in one optimizer step: mixed neighbours are scored in separate passes")
        for js in passes:

            def advantages(js=js):
                for j in js:
                    shaped[j] = self._shape([rows[j]], preps[j], rolled[j])
                return [shaped[j]["adv"] for j in js]
            last = js[-1] == acc - 1
            # every micro-batch weighs 1 / acc in the accumulated gradient (HF Trainer divides each micro-batch loss by
            # gradient_accumulation_steps): score_and_backward_multi's grad_scale is PER GROUP, whatever the pass holds
            if len(js) == 1:
                j0 = js[0]
                res = eng.score_and_backward(rolled[j0]["prompt"], rolled[j0]["completion_ids"], advantages, grad_scale=1.0 / acc,
                                             era_rule=self.era_rule, last_group=last)
            else:
                res = eng.score_and_backward_multi([rolled[j]["prompt"] for j in js], [rolled[j]["completion_ids"] for j in js],
                                                   advantages, grad_scale=1.0 / acc, era_rule=self.era_rule, last_group=last)
            for j in js:
                kls[j] = res["kl"]              # a pass's KL is the mean over its groups: the logged step mean is unchanged
                rolled[j] = None
            losses.append((len(js), res["loss"]))
        self._pending_step = (shaped, kls, losses, acc)
        return 0.0

    def _finish_step(self) -> float:
        shaped, kls, losses, acc = self._pending_step
        self._pending_step = None
        self._record_metrics(shaped, kls)
        return sum(n * float(l) for n, l in losses) / acc

    # ------------------------------------------------------------------ loop / logging / checkpoints
    def log(self, logs: Dict[str, float], start_time: Optional[float] = None) -> None:
        metrics = {k: sum(v) / len(v) for k, v in self._metrics.items()}                    # TR:689
        logs = {**logs, **metrics}
        self._metrics.clear()
        if self.rank == 0:
            os.makedirs(self.args.output_dir, exist_ok=True)
            with open(os.path.join(self.args.output_dir, "trainer_log.jsonl"), "a") as f:
                f.write(json.dumps(logs) + "\n")
            print(json.dumps(logs), flush=True)

    def train(self, resume_from_checkpoint: Optional[str] = None):
        a = self.args
        if resume_from_checkpoint:
            self._load_checkpoint(resume_from_checkpoint)
        n_rows = len(self.train_dataset)
        acc = max(1, a.gradient_accumulation_steps * a.per_device_train_batch_size)
        t_last = time.time()
        pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="spacer-prefetch")
        # resume (HF Trainer semantics): the optimizer steps already taken are skipped, not replayed -- epoch and position in
        # the epoch's permutation follow from global_step
        per_rank = (n_rows + self.world - 1) // self.world
        steps_per_epoch = max(1, (per_rank - acc) // acc + 1) if per_rank >= acc else 1
        epoch, skip = divmod(self.global_step, steps_per_epoch)

        def prep_seed(ep: int, pos: int) -> int:
            return self._sample_seed + 104729 * (ep * 1000003 + pos)

        while self.global_step < self.total_steps:
            idx = shard_indices(n_rows, self.rank, self.world, a.data_seed if a.data_seed is not None else a.seed, epoch)
            starts = list(range(0, len(idx) - acc + 1, acc))
            if not starts:
                raise ValueError(f"{len(idx)} samples per rank cannot fill one optimizer step of {acc} micro-batches "
                                 "(per_device_train_batch_size x gradient_accumulation_steps)")
            starts, skip = starts[skip:], 0

            def submit(s0: int):
                return [pool.submit(self._prepare, [self.train_dataset[idx[s0 + j]]], prep_seed(epoch, s0 + j)) for j in range(acc)]

            pending = submit(starts[0]) if starts else None
            for n, s in enumerate(starts):
                if self.global_step >= self.total_steps:
                    break
                preps = [f.result() for f in pending]
                # the next optimizer step's host work (chat template, frame decode + resize, patchify) overlaps this one's GPU work
                pending = submit(starts[n + 1]) if n + 1 < len(starts) else None
                rolled = self._rollout(preps)        # ONE decode batch for all micro-batches of the step (same weights)
                self._accumulate([self.train_dataset[idx[s + j]] for j in range(acc)], preps, rolled)
                self.engine.reduce_gradients()
                lr = self.engine.optimizer_step(self.world)
                loss = self._finish_step()           # the step's only read-back: losses, KL values, one packed metric gather
                self.global_step += 1
                if self.global_step % a.logging_steps == 0:
                    self.log({"loss": loss, "learning_rate": lr, "grad_norm": self.engine.grad_norm(self.world),
                              "step": self.global_step, "step_time": time.time() - t_last})
                    t_last = time.time()
                if a.save_steps and self.global_step % a.save_steps == 0:
                    self.save_model(os.path.join(a.output_dir, f"checkpoint-{self.global_step}"))
            if pending:
                for f in pending:
                    f.cancel()
            epoch += 1
        pool.shutdown(wait=False, cancel_futures=True)
        return {"global_step": self.global_step}

    def save_model(self, output_dir: Optional[str] = None, _internal_call: bool = False) -> None:
        """What HF ``Trainer.save_model`` leaves for the reference (open_r1/SG-RLVR.py:377-384): bf16 safetensors in the
        original Qwen2-VL / Qwen2.5-VL tensor names + config.json + tokenizer / processor files, loadable with
        ``from_pretrained`` (qwen2vl/checkpoint.py), plus the step counter for --resume_from_checkpoint.

        COLLECTIVE when ``grad_algo == "rs_ag"`` and ``save_only_model`` is false: the fp32 master weights / Adam moments live in
        the owner's shard only and are all-gathered here, so EVERY rank must call it (``train`` does); calling it on rank 0 only
        -- the HF idiom -- would hang.  With the default all-reduce exchange, or ``save_only_model=True``, nothing is communicated
        and a rank-0-only call is fine.  (Between saves under rs_ag, ``engine.master`` / ``m`` / ``v`` outside the rank's own shard
        are stale by design.)"""
        if not self.args.save_only_model:
            self.engine.gather_optimizer_state()        # collective (grad_algo rs_ag keeps master / moments per shard): every rank
        if self.rank != 0:
            return
        output_dir = output_dir or self.args.output_dir
        write_checkpoint(output_dir, self.engine.policy, source_dir=self.model_id if os.path.isdir(str(self.model_id)) else None,
                         processor=self.processing_class,
                         extra_state={"global_step": self.global_step, "model_id": self.model_id, "notes": self._log_lines})
        if not self.args.save_only_model:
            # --save_only_model false: what HF Trainer adds for an exact resume -- fp32 master weights (the sub-bf16 part of
            # the accumulated lr 1e-6 updates) and the Adam moments
            e = self.engine
            torch.save({"master": e.master.flat.cpu(), "m": e.m.cpu(), "v": e.v.cpu(), "step_count": e.step_count},
                       os.path.join(output_dir, "optimizer.pt"))

    def _load_checkpoint(self, path: str) -> None:
        load_state_dict(self.engine.policy, read_checkpoint(path))
        self.engine.master.flat.copy_(self.engine.policy.flat.float())
        st = os.path.join(path, "trainer_state.json")
        if os.path.exists(st):
            with open(st) as f:
                self.global_step = int(json.load(f).get("global_step", 0))
            self.engine.step_count = self.global_step
        opt = os.path.join(path, "optimizer.pt")
        if os.path.exists(opt):
            # mmap: the file holds 12 bytes per parameter (100 GB at 7B); every rank maps it and copies tensor by tensor instead of
            # materialising a private host copy of the whole file
            st = torch.load(opt, map_location="cpu", mmap=True)
            e = self.engine
            for name, dst in (("master", e.master.flat), ("m", e.m), ("v", e.v)):
                if st[name].numel() != dst.numel():
                    raise ValueError(f"{opt}: '{name}' has {st[name].numel()} elements, this model's flat layout has {dst.numel()} "
                                     "(checkpoint of another architecture / layout version)")
                dst.copy_(st[name])
            e.step_count = int(st["step_count"])
        self.engine.weights_changed()                  # (bumps the prefill-tape version too: every writer of the weights calls this)
