#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
{
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py tests/test_gpu_cli.py -m gpu -x -q -s -k "audio or bf16 or feature_dim" 2>&1 | grep -E "audio|passed|failed|Error|error" | head -40
rm -rf gpurun_out/atr
timeout 600 rocprofv3 --kernel-trace -d gpurun_out/atr -o a -- python scripts/audio_trace.py 32 bf16 > gpurun_out/atr_run.log 2>&1
python scripts/audio_trace.py 32 bf16 $(find gpurun_out/atr -name "a_results.db" | head -1) 2>&1 | head -40
python scripts/audio_trace.py 32 bf16 $(find gpurun_out/atr -name "a_results.db" | head -1) 2>&1 | tail -1
find gpurun_out/atr -name "*.db" -delete
} > gpurun_out/r3_audio2.log 2>&1
echo done
