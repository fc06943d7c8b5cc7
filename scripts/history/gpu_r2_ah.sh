cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16" > gpurun_out/t23.log 2>&1; echo exit=$? >> gpurun_out/t23.log; grep -a "bf16\|passed\|failed" gpurun_out/t23.log | grep -v "^tests" | tail -14 | cut -c1-200
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
timeout 300 $L > gpurun_out/ah.log 2>&1
python - <<PY
import json
s=open('gpurun_out/ah.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('bf16', d['value'], 'total ms', d['ms_per_step'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'attn' in k})
PY
grep -o '"audio[a-z_]*": {[^}]*}' gpurun_out/ah.log | head -2
