#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for rep in 1 2; do for o in unet_nb_model=0 unet_nb_model=1; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_roofline --no_secondary --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline $o', d['value'], d['ms_per_step'])"
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or cfg_1s or step_counts or reference_own" 2>&1 | tail -3
} > gpurun_out/r3_nb2.log 2>&1
echo done
