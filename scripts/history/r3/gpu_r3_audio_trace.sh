#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/atr
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/atr -o a -- python scripts/audio_trace.py 32 bf16 > gpurun_out/atr_run.log 2>&1
python scripts/audio_trace.py 32 bf16 $(find gpurun_out/atr -name "a_results.db" | head -1) > gpurun_out/r3_audio_trace.txt 2>&1
find gpurun_out/atr -name "*.db" -delete
echo done
