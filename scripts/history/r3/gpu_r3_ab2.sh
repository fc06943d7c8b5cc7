#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python scripts/option_ab.py f32_out1_tm fp32 32 50
timeout 600 python scripts/option_ab.py hybrid_f32 fp32 32 50
echo "== GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 timeout 600 python scripts/clip_groups_sweep.py 600 2>&1 | grep -E "B= 32|B= 64|B= 16"
} > gpurun_out/r3_ab2.log 2>&1
echo done
