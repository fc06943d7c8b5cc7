cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "unet_forward or audio_encoder_vs_golden or large_batch" > gpurun_out/t22.log 2>&1; echo exit=$? >> gpurun_out/t22.log; tail -3 gpurun_out/t22.log | cut -c1-200
timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/ag1.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|gathered_checksum": [0-9.]*' gpurun_out/ag1.log | tr '\n' ' '; echo " <- B=1"
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 > gpurun_out/ag2.log 2>&1
python - <<PY
import json
s=open('gpurun_out/ag2.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('f32 B=32', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'attn' in k})
PY
