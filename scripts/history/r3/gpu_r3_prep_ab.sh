#!/bin/bash
# same-box A/B of two engine builds (said_amd/lib/ab_old.so / ab_new.so) + the large-batch parity tests on the new one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for rep in 1 2; do for l in old new; do timeout 300 python scripts/ab_libs.py said_amd/lib/ab_$l.so 32 50; done; done
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -k "batch32 or token_major or large_batch or clip_groups or batch_driver or bf16" 2>&1 | tail -5
} > gpurun_out/r3_prep_ab.log 2>&1
echo done
