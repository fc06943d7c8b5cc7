#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for o in unet_nb_model=0 unet_nb_model=1 unet_nb=1 unet_nb=3; do
  timeout 300 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 $o', d['value'], d['ms_per_step'])"
done
for o in unet_nb_model=0 unet_nb_model=1; do
  timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline $o', d['value'], d['ms_per_step'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "long_sequence or editing or batch32 or ragged or single_clip or cfg_1s or step_counts" 2>&1 | tail -3
} > gpurun_out/r3_nb.log 2>&1
echo done
