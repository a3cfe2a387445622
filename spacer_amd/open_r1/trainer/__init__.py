"""Same import surface as the reference's ``open_r1/trainer/__init__.py``: ``from trainer import SGRLVRTrainer``."""
from .SG_RLVR_trainer import SGRLVRTrainer
from .vllm_grpo_trainer_modified import Qwen2VLGRPOVLLMTrainerModified

# The reference also ships Qwen2VLGRPOTrainer (grpo_trainer.py): byte-for-byte the same compute_loss except that it does
# not pass video_path to the reward functions.  Kept as an alias of the same step.
Qwen2VLGRPOTrainer = SGRLVRTrainer

__all__ = ["Qwen2VLGRPOTrainer", "Qwen2VLGRPOVLLMTrainerModified", "SGRLVRTrainer"]
