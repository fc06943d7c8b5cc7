cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# GroupNorm coefficients finalised inside the preparation kernel (vs their own launch); preparation-kernel occupancy
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "large_batch or batch32" > gpurun_out/t17.log 2>&1; echo exit=$? >> gpurun_out/t17.log; grep -a "max err\|passed\|failed\|Error\|error" gpurun_out/t17.log | tail -9 | cut -c1-300
for dt in bf16 f32; do
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype $dt"
for v in fused separate pad24k pad48k; do
unset SAID_PREP_GN_SEPARATE SAID_PREP_PAD_LDS
[ $v = separate ] && export SAID_PREP_GN_SEPARATE=1
[ $v = pad24k ] && export SAID_PREP_PAD_LDS=24576
[ $v = pad48k ] && export SAID_PREP_PAD_LDS=49152
timeout 300 $L > gpurun_out/y.log 2>&1
python - <<PY
import json
s=open('gpurun_out/y.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('$dt $v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], r['unet_step']['launches'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'prep' in k})
PY
done; done
