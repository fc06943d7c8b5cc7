#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for o in mt_wgs=0 mt_wgs=228 mt_wgs=342 mt_wgs=0 mt_wgs=228; do
  timeout 300 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 $o', d['value'], d['ms_per_step'])"
done
timeout 600 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s -k "clone_workspace" 2>&1 | grep -E "clone growth|passed|failed|Error"
} > gpurun_out/r3_mt.log 2>&1
echo done
