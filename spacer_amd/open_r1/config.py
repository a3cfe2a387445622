"""Flag surface of the launch script (run_SpaceR_SG_RLVR.sh:15-39): the three dataclasses the reference parses with
``TrlParser((GRPOScriptArguments, GRPOConfig, ModelConfig))`` (SG-RLVR.py:389-392), restated without trl / HF Trainer.
Every flag of the shipped script is accepted; the ones that configure machinery this engine does not have
(--deepspeed, --attn_implementation, --report_to) are accepted and ignored with a log line; --gradient_checkpointing selects
the engine's selective activation recompute.
"""
from __future__ import annotations

import argparse
from dataclasses import MISSING, dataclass, field, fields
from typing import List, Optional


def _bool(v) -> bool:
    if isinstance(v, bool):
        return v
    return str(v).strip().lower() in ("1", "true", "yes", "y", "t")


@dataclass
class GRPOScriptArguments:                     # SG-RLVR.py:27-56
    dataset_name: str = ""
    dataset_config: Optional[str] = None
    dataset_train_split: str = "train"
    dataset_test_split: str = "test"
    reward_funcs: List[str] = field(default_factory=lambda: ["accuracy", "format"])
    max_pixels: Optional[int] = 12845056
    min_pixels: Optional[int] = 3136
    temporal: Optional[bool] = False
    len_control: Optional[bool] = True
    map_annotation: str = "annotation/cognitive_map.jsonl"      # path the reference hard-codes (:291)
    # M-RoPE text position after a vision span: True = the rule of the transformers 4.49-dev build the reference pins
    # (r1-v/setup.py:64): max(all vision positions) + 1; False = transformers 5.x: start + max(h, w) / merge.  They differ
    # when the temporal extent exceeds the spatial one (Qwen2.5-VL videos: 16 frames, tokens_per_second 2 -> t up to s + 14
    # against 11..14 merged rows / columns), so checkpoints must be trained with the rule their consumers use.
    mrope_era_rule: Optional[bool] = True


@dataclass
class GRPOConfig:                              # the trl.GRPOConfig / TrainingArguments fields the path reads
    output_dir: str = "./log/SpaceR"
    max_prompt_length: Optional[int] = 16384
    max_completion_length: int = 1024
    num_generations: int = 8
    beta: float = 0.04
    per_device_train_batch_size: int = 1
    gradient_accumulation_steps: int = 1
    learning_rate: float = 1e-6
    lr_scheduler_type: str = "cosine"
    weight_decay: float = 0.01
    warmup_steps: int = 0
    adam_beta1: float = 0.9
    adam_beta2: float = 0.999
    adam_epsilon: float = 1e-8
    max_grad_norm: float = 5.0
    num_train_epochs: float = 1.0
    max_steps: int = -1
    logging_steps: int = 1
    save_steps: int = 1000
    save_only_model: bool = True
    bf16: bool = True
    gradient_checkpointing: bool = False       # HF TrainingArguments default; the shipped script passes true (SC:28) -> GRPOHyper.recompute
    seed: int = 42
    data_seed: Optional[int] = None
    resume_from_checkpoint: Optional[str] = None
    run_name: Optional[str] = None
    report_to: Optional[str] = None
    deepspeed: Optional[str] = None
    top_k: int = 50                # era default of the GenerationConfig the reference builds
    use_decode_graph: bool = True
    # gradient-accumulation micro-batches scored / back-propagated per token-packed pass (GRPOEngine.score_and_backward_multi);
    # 2 fits 288 GB at 7B with 16-frame prompts and 512-token rollouts, 1 = one prompt group per pass
    groups_per_pass: int = 2
    # --precise_logps true: policy and reference per-token log-probs (KL, loss, metrics: TR:527-552) evaluated in the precise mode
    # (within 1e-3 of an fp32 evaluation at full 7B depth; GRPOHyper.precise_logps); the gradient runs on the production backward
    precise_logps: bool = False
    grad_algo: str = "allreduce"               # data-parallel exchange: "allreduce" (overlapped, replicated AdamW) or "rs_ag" (GRPOHyper.grad_algo)
    reuse_prefill: str = "auto"                # the rollout's prefill keeps its tape for the policy's scoring pass: auto (when it fits) | true | false (GRPOHyper.reuse_prefill)
    # vLLM-trainer topology (trl.GRPOConfig fields read by vllm_grpo_trainer_modified.py:300,325,362): generation on a dedicated GPU
    use_vllm: bool = False
    vllm_device: str = "auto"                  # "auto" = the first GPU index after the training ranks (:325-327)
    vllm_gpu_memory_utilization: float = 0.9   # accepted and ignored


@dataclass
class ModelConfig:
    model_name_or_path: str = ""
    attn_implementation: Optional[str] = "flash_attention_2"
    torch_dtype: Optional[str] = "bfloat16"


def parse_args(argv=None):
    """``--flag value`` parser over the three dataclasses (TrlParser.parse_args_and_config equivalent)."""
    ap = argparse.ArgumentParser(description="SG-RLVR on the MI355X-native engine")
    classes = (GRPOScriptArguments, GRPOConfig, ModelConfig)
    for cls in classes:
        for f in fields(cls):
            default = f.default if f.default is not MISSING else f.default_factory()  # type: ignore
            ann = str(f.type)
            if "bool" in ann:
                ap.add_argument(f"--{f.name}", type=_bool, default=default, nargs="?", const=True)
            elif "List" in ann:
                ap.add_argument(f"--{f.name}", nargs="+", default=default)
            elif "int" in ann:
                ap.add_argument(f"--{f.name}", type=int, default=default)
            elif "float" in ann:
                ap.add_argument(f"--{f.name}", type=float, default=default)
            else:
                ap.add_argument(f"--{f.name}", type=str, default=default)
    ns = ap.parse_args(argv)
    out = []
    for cls in classes:
        out.append(cls(**{f.name: getattr(ns, f.name) for f in fields(cls)}))
    return tuple(out)
