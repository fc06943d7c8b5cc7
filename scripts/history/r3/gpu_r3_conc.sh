#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s -k "clip_groups" 2>&1 | grep -E "clip groups|clone growth|passed|failed|Error"
for rep in 1 2; do
timeout 600 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32 B=32 N=100', d['value'], d['ms_per_step'])"
done
} > gpurun_out/r3_conc.log 2>&1
echo done
