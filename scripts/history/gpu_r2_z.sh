cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# epilogue split between the two K-half waves of the 64-row tile; ragged-length test of the token-major path
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "large_batch or batch32 or ragged_length" > gpurun_out/t18.log 2>&1; echo exit=$? >> gpurun_out/t18.log; grep -a "sample\|passed\|failed\|Error\|error" gpurun_out/t18.log | tail -22 | cut -c1-200
for dt in bf16 f32; do
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype $dt > gpurun_out/z.log 2>&1
python - <<PY
import json
s=open('gpurun_out/z.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('$dt', d['value'], 'step', r['unet_step']['ms_loop_per_step'], r['unet_step']['launches'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'gemm_kernel<' in k and ('tgemm' in k or 'fgemm' in k)})
PY
done
