cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# fp32 token-major GEMM, small-workgroup shape (fgemm2_kernel) against the 4- / 8-wave shapes
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "large_batch or batch32 or long_sequence" > gpurun_out/t14.log 2>&1; echo exit=$? >> gpurun_out/t14.log; grep -a "max err\|passed\|failed\|Error\|error" gpurun_out/t14.log | tail -8 | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for v in 0 1 2 3; do
SAID_FGEMM_V1=$v timeout 300 $L > gpurun_out/o_v$v.log 2>&1
python - <<PY
import json
s=open('gpurun_out/o_v$v.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('SAID_FGEMM_V1=$v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
done
find gpurun_out -name "*.db" -delete
