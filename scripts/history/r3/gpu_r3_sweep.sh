#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{ timeout 1500 python scripts/clip_groups_sweep.py 600; timeout 600 python scripts/clip_groups_sweep.py 1800; } > gpurun_out/r3_sweep.log 2>&1
echo done
