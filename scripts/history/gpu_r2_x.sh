cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# bf16 mode: the small K-split tile (fgemm_kernel<.., BF>) against the 256-row tiles, N % 96 shapes (bit 0) and GEGLU (bit 1)
SAID_TGEMM_SMALL=3 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16_unet_large_batch or bf16_loop_cfg_batch32" > gpurun_out/t16.log 2>&1; echo exit=$? >> gpurun_out/t16.log; grep -a "max err\|passed\|failed\|Error\|error" gpurun_out/t16.log | tail -6 | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
for v in 0 1 2 3; do
SAID_TGEMM_SMALL=$v timeout 300 $L > gpurun_out/x_v$v.log 2>&1
python - <<PY
import json
s=open('gpurun_out/x_v$v.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('SAID_TGEMM_SMALL=$v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
done
bash scripts/gpu_r2_w.sh
