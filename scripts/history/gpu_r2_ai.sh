cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "large_batch or ragged_length or batch32" > gpurun_out/t24.log 2>&1; echo exit=$? >> gpurun_out/t24.log; tail -3 gpurun_out/t24.log | cut -c1-200
for dt in bf16 f32; do
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype $dt > gpurun_out/ai.log 2>&1
python - <<PY
import json
s=open('gpurun_out/ai.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('$dt', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'prep' in k})
PY
done
