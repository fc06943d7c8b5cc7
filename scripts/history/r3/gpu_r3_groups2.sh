#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
bash scripts/gpu_r3_groups.sh; cat gpurun_out/r3_groups.log
timeout 1500 python scripts/clip_groups_sweep.py 600
} > gpurun_out/r3_groups2.log 2>&1
echo done
