#!/bin/bash
# clip groups: parity tests, then bench with and without the split on the same box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 900 python -m pytest tests/test_gpu_round3.py -m gpu -x -q -s -k "clip_groups or eta" 2>&1 | tail -15
for g in 0 2; do
  echo "== --clip_groups $g"
  timeout 900 python bench.py --steps 2 --warmup 1 --clip_groups $g 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'])
for k,v in d.get('secondary',{}).items(): print(k, {kk:v[kk] for kk in v if kk in ('value','ms_per_step','ms_per_denoise_step','clip_groups')})
"
done
} > gpurun_out/r3_groups.log 2>&1
echo done
