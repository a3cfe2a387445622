#!/bin/bash
# Drop-in for SpaceR-SG-RLVR/src/scripts/run_SpaceR_SG_RLVR.sh: same flags, one process per MI355X over RCCL/xGMI.
export DEBUG_MODE="${DEBUG_MODE:-false}"   # "true": append every rollout + reward to LOG_PATH (reference default: false, SC:3)
export LOG_PATH="./debug_log_SpaceR.txt"
export HSA_ENABLE_IPC_MODE_LEGACY=0
# For resume training:  --resume_from_checkpoint Model_Path \
# Set temporal to false to speed training (the T-GRPO branch adds a second rollout with K/2 completions)

torchrun --nproc_per_node="${NPROC:-8}" --nnodes="1" --node_rank="0" \
    --master_addr="127.0.0.1" --master_port="12365" \
    -m spacer_amd.open_r1.SG_RLVR \
    --output_dir "./log/SpaceR" \
    --model_name_or_path "${MODEL:-Qwen/Qwen2.5-VL-7B-Instruct}" \
    --dataset_name "${DATASET:-SpaceR-151k.jsonl}" \
    --deepspeed local_scripts/zero3.json \
    --temporal true \
    --len_control true \
    --max_prompt_length 16384 \
    --max_completion_length 1024 \
    --per_device_train_batch_size 1 \
    --gradient_accumulation_steps 1 \
    --learning_rate 1e-6 \
    --lr_scheduler_type "cosine" \
    --weight_decay 0.01 \
    --bf16 \
    --logging_steps 1 \
    --gradient_checkpointing true \
    --attn_implementation flash_attention_2 \
    --max_pixels 401408 \
    --num_train_epochs 1 \
    --run_name SpaceR \
    --save_steps 1000 \
    --beta 0.04 \
    --max_grad_norm 5 \
    --save_only_model true \
    --num_generations 8
