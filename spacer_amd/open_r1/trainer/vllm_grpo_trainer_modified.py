"""``Qwen2VLGRPOVLLMTrainerModified`` of the reference (open_r1/trainer/vllm_grpo_trainer_modified.py:83) on this engine: the
same SG-RLVR step as ``SGRLVRTrainer`` with generation moved to a dedicated rollout GPU.

Reference topology: N training processes + the main process driving a vLLM engine on GPU index N (:317-391); per step a
weight reload (:526-545), a gather of all prompts (:548-549), one ``llm.generate`` with ``n = num_generations`` and prefix
caching (:565-593) and a broadcast of the completion ids (:604-609).  Here the rollout engine is the job's LAST RANK
(`spacer_amd/rollout_server.py`: sharded weight push over all xGMI links, point-to-point prompt transfer, one decode loop
for the whole job's rows, scatter of the ids); the training ranks run everything else of ``compute_loss`` unchanged.

Launch: one process more than training GPUs (the reference: one GPU more than ``--num_processes``), ``--use_vllm true``;
`open_r1/SG_RLVR.py:main` sends the last rank into ``run_rollout_rank`` and builds this trainer on the others.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ...qwen2vl.config import Qwen2VLConfig
from ...rollout import PromptInput, SamplingParams
from ...rollout_server import RolloutClient, RolloutServer, RolloutTopology
from .SG_RLVR_trainer import SGRLVRTrainer


class Qwen2VLGRPOVLLMTrainerModified(SGRLVRTrainer):
    def __init__(self, *args, topology: RolloutTopology, **kwargs):
        if topology.is_server:
            raise ValueError("the rollout rank does not build a trainer: call run_rollout_rank() there")
        kwargs["process_group"] = topology.trainer_pg if topology.n_trainers > 1 else None
        super().__init__(*args, **kwargs)
        if not getattr(self.args, "use_vllm", True):
            # vllm_grpo_trainer_modified.py:386-389: this trainer only knows the dedicated-device generation path
            raise ValueError("GRPOVLLMTrainerModified only supports vllm generation, please set --use_vllm True")
        dev = getattr(self.args, "vllm_device", "auto")
        if dev not in ("auto", f"cuda:{topology.server_rank}"):
            raise ValueError(f"--vllm_device {dev}: the rollout engine is the job's last rank (GPU index {topology.server_rank}, the "
                             "reference's 'auto' rule); other placements are not supported")
        self.topology = topology
        self.client = RolloutClient(topology, self.engine.policy.flat)
        self._note(f"generation on the dedicated rollout rank {topology.server_rank}; --vllm_gpu_memory_utilization accepted and "
                   "ignored (the rollout rank holds one bf16 policy copy + KV caches)")

    def _generate(self, prompts: List[PromptInput], n, sp: SamplingParams) -> torch.Tensor:
        # weights_version = global_step: pushed once per optimizer step, not per accumulation micro-step (:526, :545)
        if isinstance(n, int):
            return self.client.generate(prompts, n, sp, weights_version=self.global_step)
        # per-prompt counts (the twins' G // 2): the rollout rank's protocol carries ONE count -- generate max(n) everywhere and keep the
        # first n[i] rollouts of prompt i
        top = max(n)
        ids = self.client.generate(prompts, top, sp, weights_version=self.global_step)
        keep = torch.cat([torch.arange(i * top, i * top + k) for i, k in enumerate(n)]).to(ids.device)
        return ids.index_select(0, keep)

    def train(self, resume_from_checkpoint: Optional[str] = None):
        try:
            return super().train(resume_from_checkpoint)
        finally:
            self.client.shutdown()


def run_rollout_rank(cfg: Qwen2VLConfig, topology: RolloutTopology, device) -> int:
    """Body of the rollout rank: an empty bf16 parameter buffer (filled by the first weight push -- the checkpoint is read by
    the training ranks only), the forward engine + RolloutEngine on top of it, then serve until the trainers shut down.
    Returns the number of generate requests served."""
    from ...qwen2vl.engine import Qwen2VLEngine
    from ...qwen2vl.weights import FlatParams
    from ...rollout import RolloutEngine
    params = FlatParams.empty(cfg, device)
    roll = RolloutEngine(Qwen2VLEngine(cfg, params))
    return RolloutServer(topology, params.flat, roll, device=device).serve()
