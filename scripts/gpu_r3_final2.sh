#!/bin/bash
# evidence batch: scripts/gpu_r3_final.sh (suite, smoke, default bench line, kernel traces, SQ counters, schedule A/B) followed by the
# PMC traffic passes (scripts/gpu_r3_traffic.sh) on the SAME snapshot
cd "$GRAFT_REPO_ROOT"
bash scripts/gpu_r3_final.sh
bash scripts/gpu_r3_traffic.sh > gpurun_out/final/traffic_run.log 2>&1
tail -3 gpurun_out/final/traffic_run.log
