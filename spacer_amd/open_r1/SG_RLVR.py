"""Entry script with the reference's flag surface (open_r1/SG-RLVR.py:260-392; a '-' cannot appear in a module name):

    torchrun --nproc_per_node=8 -m spacer_amd.open_r1.SG_RLVR --model_name_or_path <ckpt dir> --dataset_name <jsonl> ...

Builds the prompts exactly as the reference (COGMAP / QUESTION / TYPE templates, :293-352), loads the cognitive-map
annotations into the reward module, constructs SGRLVRTrainer and trains.
"""
from __future__ import annotations

import json
import os

import torch

from . import rewards as R
from .config import parse_args
from .trainer import Qwen2VLGRPOVLLMTrainerModified, SGRLVRTrainer

EXAMPLE_MAP = {"table": [[0, 3], [5, 7]], "chair": [[9, 3]], "window": [[6, 5]]}
_THINK = ("Please think about this question as if you were a human pondering deeply. "
          "Engage in an internal dialogue using expressions such as 'let me think', 'wait', 'Hmm', 'oh, I see', 'let's break it down', etc, "
          "or other natural language thought expressions It's encouraged to include self-reflection or verification in the reasoning process")
QUESTION_TEMPLATE = ("Question: {Question}\n" + _THINK + ". Provide your detailed reasoning between the <think> </think> tags, and then give "
                     "your final answer between the <answer> </answer> tags.")
COGMAP_TEMPLATE = (
    "Question: {Question}\n" + _THINK + ".\n"
    "If generating a cognitive map for the video can help you answer the question, you could follow the below steps to generate a "
    "cognitive map in <map> </map> tags\n"
    "[Steps] Identify specific objects within the **video scene**, understand the spatial arrangement of the scene, and estimate the "
    "center point of each object, assuming the entire scene is represented by a 10x10 grid. These information should be summarized in "
    "<map> </map> tags.\n"
    "[Rule]1. We provide the categories to care about in this scene: {object_list}. Focus ONLY on these categories for the entire "
    "video scene.\n2. Estimate the center location of each instance within the provided categories, assuming the entire scene is "
    "represented by a 10x10 grid, considering the information from all frames.\n3. If a category contains multiple instances across "
    "all frames, include all of them.\n"
    "Present the map using dict format. Here is an example: <map>{map_example}</map>.\n"
    "If you generate a cognitive map, please put it in <map> </map> tags. Provide your detailed reasoning process between the <think> "
    "</think> tags, and then give your final answer between the <answer> </answer> tags.")
TYPE_TEMPLATE = {
    "multiple choice": " Please provide only the single option letter (e.g., A, B, C, D, etc.) within the <answer> </answer> tags.",
    "numerical": " Please provide the numerical value (e.g., 42 or 3.1) within the <answer> </answer> tags.",
    "OCR": " Please transcribe text from the image/video clearly and provide your text answer within the <answer> </answer> tags.",
    "free-form": " Please provide your text answer within the <answer> </answer> tags.",
    "regression": " Please provide the numerical value (e.g., 42 or 3.14) within the <answer> </answer> tags.",
}


def make_conversation_image_and_video_map(example: dict) -> dict:
    """SG-RLVR.py:319-352: one user turn = [media part, text part]."""
    question = example["problem"]
    if example["problem_type"] == "multiple choice":
        question += "Options:\n" + "".join(op + "\n" for op in example["options"])
    if example.get("data_source") == "SR_dataset":
        vid = os.path.splitext(os.path.basename(example["path"]))[0]
        objs = list(R.MAP_DATA[vid]["cognitive_map"].keys())
        text = COGMAP_TEMPLATE.format(Question=question, object_list=objs, map_example=EXAMPLE_MAP)
    else:
        text = QUESTION_TEMPLATE.format(Question=question)
    text += TYPE_TEMPLATE[example["problem_type"]]
    return {"prompt": [{"role": "user", "content": [{"type": example["data_type"]}, {"type": "text", "text": text}]}]}


def load_rows(path: str):
    with open(path, "r", encoding="utf-8") as f:
        if path.endswith(".jsonl"):
            return [json.loads(line) for line in f if line.strip()]
        return json.load(f)


def main(script_args, training_args, model_args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    pg = None
    if world > 1:
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        torch.distributed.init_process_group("nccl")           # RCCL over xGMI
        pg = torch.distributed.group.WORLD
    topology = None
    if training_args.use_vllm:                  # N training ranks + the last rank as the dedicated rollout GPU
        if pg is None:
            raise ValueError("--use_vllm true needs one process more than training GPUs (torchrun --nproc_per_node N+1): the last "
                             "rank is the rollout engine, as the reference needs one GPU more than --num_processes")
        from ..rollout_server import make_topology
        from .trainer.vllm_grpo_trainer_modified import run_rollout_rank
        topology = make_topology(pg)
        if topology.is_server:
            from ..qwen2vl.checkpoint import config_of_dir
            from ..qwen2vl.config import preset_for
            cfg = config_of_dir(model_args.model_name_or_path) or preset_for(model_args.model_name_or_path)
            served = run_rollout_rank(cfg, topology, torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
            print(f"[spacer_amd] rollout rank served {served} generate requests", flush=True)
            return
    reward_funcs = [R.reward_funcs_registry[n] for n in script_args.reward_funcs]
    if os.path.exists(script_args.map_annotation):
        R.load_map(script_args.map_annotation)
    rows = load_rows(script_args.dataset_name)
    rows = [{**r, **make_conversation_image_and_video_map(r)} for r in rows]
    from transformers import AutoProcessor                      # the reference's processing_class (TR:226)
    processor = AutoProcessor.from_pretrained(model_args.model_name_or_path)
    processor.pad_token_id = processor.tokenizer.pad_token_id
    processor.eos_token_id = processor.tokenizer.eos_token_id
    common = dict(model=model_args.model_name_or_path, reward_funcs=reward_funcs, args=training_args, script_args=script_args,
                  train_dataset=rows, processing_class=processor, attn_implementation=model_args.attn_implementation,
                  max_pixels=script_args.max_pixels, min_pixels=script_args.min_pixels)
    if topology is not None:
        trainer = Qwen2VLGRPOVLLMTrainerModified(topology=topology, **common)
    else:
        trainer = SGRLVRTrainer(process_group=pg, **common)          # SG-RLVR.py:360 hard-codes this class
    trainer.train(resume_from_checkpoint=training_args.resume_from_checkpoint)
    trainer.save_model(training_args.output_dir)


if __name__ == "__main__":
    main(*parse_args())
