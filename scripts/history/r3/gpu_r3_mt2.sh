#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for rep in 1 2; do for o in mt_mid=0 mt_mid=1; do
  timeout 300 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('cfg4 $o', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --steps 4 --warmup 2 --no_cpu_baseline --no_roofline --no_secondary --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline $o', d['value'], d['ms_per_step'])"
  timeout 300 python bench.py --batch 4 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --debug_option $o 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('B=4 T=600 $o', d['value'], d['ms_per_step'])"
done; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -x -q -k "long_sequence or editing or ragged or single_clip or cfg_1s or step_counts or golden or batch32 or token_major" 2>&1 | tail -3
} > gpurun_out/r3_mt2.log 2>&1
echo done
