"""Supervised fine-tuning entry on the same engine -- the counterpart of the reference's ``open_r1/sft.py``:
same conversation builder, same label rule (pad and visual placeholder tokens are ignored, sft.py:170-181), the HF causal-LM
loss (mean cross-entropy of the shifted labels) computed and back-propagated by libspacer_hip's kernels through
``GRPOEngine.sft_forward_backward``; AdamW / clipping / DP all-reduce are the ones of the GRPO path.

    python -m spacer_amd.open_r1.sft --model_name_or_path <dir> --dataset_name data.jsonl --output_dir out ...
"""
from __future__ import annotations

import json
import os
import sys
import time
from typing import Any, Dict, List

import torch

from ..grpo import GRPOEngine, GRPOHyper
from ..qwen2vl.checkpoint import config_of_dir, read_checkpoint, write_checkpoint
from ..qwen2vl.config import preset_for
from ..qwen2vl.weights import FlatParams, load_state_dict

from .SG_RLVR import QUESTION_TEMPLATE as _GRPO_QUESTION_TEMPLATE
from .SG_RLVR import TYPE_TEMPLATE as _GRPO_TYPE_TEMPLATE

SYSTEM_MESSAGE = "You are a helpful assistant"                                            # sft.py:89
# sft.py:92-98: the GRPO prompt (open_r1/SG-RLVR.py, reproduced verbatim in SG_RLVR.py) without its "Question: " label
QUESTION_TEMPLATE = _GRPO_QUESTION_TEMPLATE.replace("Question: {Question}", "{Question}", 1)
# sft.py:100-106: the GRPO answer-format hints, except that the numeric example reads 3.14 here (3.1 in SG-RLVR.py)
TYPE_TEMPLATE = {k: v.replace("(e.g., 42 or 3.1)", "(e.g., 42 or 3.14)") for k, v in _GRPO_TYPE_TEMPLATE.items()}


def prepare_dataset(example: Dict[str, Any]) -> Dict[str, Any]:
    """One dataset row -> system / user (media + question + answer-format hint) / assistant (solution) turns
    (reference ``prepare_dataset``, sft.py:84-143; outputs pinned by tests/golden/sft_conversations.json)."""
    if example["problem_type"] == "multiple choice":
        question = example["problem"] + "Options:\n" + "".join(op + "\n" for op in example["options"])
    else:
        question = example["problem"]
    return {"messages": [
        {"role": "system", "content": [{"type": "text", "text": SYSTEM_MESSAGE}]},
        {"role": "user", "content": [
            {"type": example["data_type"], example["data_type"]: example["path"]},
            {"type": "text", "text": QUESTION_TEMPLATE.format(Question=question) + TYPE_TEMPLATE[example["problem_type"]]}]},
        {"role": "assistant", "content": [{"type": "text", "text": example["solution"]}]},
    ]}


make_conversation = prepare_dataset


def label_mask(input_ids: torch.Tensor, pad_token_id: int, visual_token_ids) -> torch.Tensor:
    """True where the reference keeps the label (labels != -100): everything except pad and visual tokens."""
    keep = input_ids != pad_token_id
    for v in visual_token_ids:
        keep &= input_ids != v
    return keep


def collate(example: Dict[str, Any], processor, cfg, device):
    """One conversation -> (ids [S], pix, grids, keep-mask [S], second_per_grid_ts); batch size 1 per call like the GRPO path."""
    from ..qwen_vl_utils.vision_process import process_vision_info
    msgs = example["messages"]
    text = processor.apply_chat_template(msgs, tokenize=False)
    image_inputs, video_inputs, _ = process_vision_info(msgs, return_video_kwargs=True)
    out = processor(text=[text], images=image_inputs, videos=video_inputs, return_tensors="pt", padding=True)
    ids = out["input_ids"][0].to(device).long()
    pix, grids = None, None
    key = "pixel_values_videos" if "pixel_values_videos" in out else ("pixel_values" if "pixel_values" in out else None)
    if key is not None:
        pv = out[key].to(device)
        pix = torch.zeros(pv.shape[0], cfg.patch_kpad, device=device, dtype=torch.bfloat16)
        pix[:, :pv.shape[1]] = pv.to(torch.bfloat16)
        g = out["video_grid_thw" if key == "pixel_values_videos" else "image_grid_thw"]
        grids = [tuple(int(v) for v in row) for row in g.tolist()]
    pad = getattr(getattr(processor, "tokenizer", processor), "pad_token_id", cfg.pad_token_id)
    keep = label_mask(ids, pad, (cfg.vision_start_id, cfg.vision_end_id, cfg.video_token_id, cfg.image_token_id))
    sec = out.get("second_per_grid_ts") if hasattr(out, "get") else None
    if sec is not None:
        sec = [float(v) for v in (sec.tolist() if hasattr(sec, "tolist") else sec)]
    return ids, pix, grids, keep, sec


def train(model_path: str, rows: List[Dict[str, Any]], processor, *, output_dir: str, learning_rate: float = 1e-5,
          epochs: int = 1, grad_accum: int = 1, max_grad_norm: float = 1.0, weight_decay: float = 0.0, device=None,
          process_group=None, log_every: int = 1) -> GRPOEngine:
    device = device or torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    cfg = config_of_dir(model_path) or preset_for(model_path)
    params = FlatParams.empty(cfg, device)
    load_state_dict(params, read_checkpoint(model_path))
    world = torch.distributed.get_world_size(process_group) if process_group is not None else 1
    rank = torch.distributed.get_rank(process_group) if process_group is not None else 0
    # equal-length, padded shards (DistributedSampler semantics, the GRPO trainer's shard_indices): every rank takes the
    # same number of optimizer steps and derives the same linear schedule from the GLOBAL row count
    from .trainer.SG_RLVR_trainer import shard_indices
    mine = [rows[i] for i in shard_indices(len(rows), rank, world, seed=0, epoch=0, shuffle=False)]
    steps = max(1, (len(mine) // grad_accum) * epochs)
    hyper = GRPOHyper(learning_rate=learning_rate, weight_decay=weight_decay, max_grad_norm=max_grad_norm, total_steps=steps,
                      lr_scheduler_type="linear")
    eng = GRPOEngine(cfg, params, hyper, ref=params, process_group=process_group)     # no frozen reference model in SFT
    step, t0 = 0, time.time()
    for ep in range(epochs):
        for s in range(0, len(mine) - grad_accum + 1, grad_accum):
            loss = 0.0
            for j in range(grad_accum):
                ids, pix, grids, keep, sec = collate(make_conversation(mine[s + j]) if "messages" not in mine[s + j] else mine[s + j],
                                                     processor, cfg, device)
                loss += eng.sft_forward_backward(ids, pix, grids, keep, grad_scale=1.0 / grad_accum, second_per_grid_ts=sec,
                                                 last_group=j == grad_accum - 1) / grad_accum
            eng.reduce_gradients()
            lr = eng.optimizer_step(world)
            step += 1
            if rank == 0 and step % log_every == 0:
                print(json.dumps({"step": step, "loss": loss, "learning_rate": lr, "step_time": time.time() - t0}), flush=True)
                t0 = time.time()
    if rank == 0:
        write_checkpoint(output_dir, eng.policy, source_dir=model_path if os.path.isdir(model_path) else None, processor=processor,
                         extra_state={"global_step": step, "objective": "sft"})
    return eng


def main(argv=None) -> None:
    import argparse
    ap = argparse.ArgumentParser(description="SFT on libspacer_hip (reference: open_r1/sft.py)")
    ap.add_argument("--model_name_or_path", required=True)
    ap.add_argument("--dataset_name", required=True, help=".json / .jsonl of SpaceR rows (problem, options, solution, path, data_type, problem_type)")
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--learning_rate", type=float, default=1e-5)
    ap.add_argument("--num_train_epochs", type=int, default=1)
    ap.add_argument("--gradient_accumulation_steps", type=int, default=1)
    ap.add_argument("--max_grad_norm", type=float, default=1.0)
    ap.add_argument("--weight_decay", type=float, default=0.0)
    a = ap.parse_args(argv)
    from transformers import AutoProcessor
    processor = AutoProcessor.from_pretrained(a.model_name_or_path)
    with open(a.dataset_name) as f:
        rows = json.load(f) if a.dataset_name.endswith(".json") else [json.loads(l) for l in f if l.strip()]
    pg = None
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)                       # before the RCCL communicator is created
        torch.distributed.init_process_group("nccl", device_id=dev)
        pg = torch.distributed.group.WORLD
    train(a.model_name_or_path, rows, processor, output_dir=a.output_dir, learning_rate=a.learning_rate, epochs=a.num_train_epochs,
          grad_accum=a.gradient_accumulation_steps, max_grad_norm=a.max_grad_norm, weight_decay=a.weight_decay, process_group=pg)


if __name__ == "__main__":
    main(sys.argv[1:])
