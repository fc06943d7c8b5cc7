#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -m gpu -x -q -s -k "audio or bf16 or clip_groups or batch32" 2>&1 | tail -25
timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['roofline']['audio_encode'])"
} > gpurun_out/r3_audio.log 2>&1
echo done
