#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python scripts/option_ab.py f32_out1_tm fp32 32 50
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "batch32 or token_major or large_batch or clip_groups or batch_driver or fp32_unet or ragged" 2>&1 | tail -5
} > gpurun_out/r3_out1.log 2>&1
echo done
