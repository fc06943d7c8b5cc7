cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for v in 0 8; do
SAID_FG_KO=$v timeout 300 $L > gpurun_out/u_ko$v.log 2>&1
python - <<PY
import json
s=open('gpurun_out/u_ko$v.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('fgemm3 variant $v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k})
PY
done
