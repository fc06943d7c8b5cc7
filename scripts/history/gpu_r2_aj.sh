cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
for v in default 3 0; do
if [ $v = default ]; then unset SAID_TGEMM_SMALL; else export SAID_TGEMM_SMALL=$v; fi
timeout 300 $L > gpurun_out/aj.log 2>&1
python - <<PY
import json
s=open('gpurun_out/aj.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('SAID_TGEMM_SMALL=$v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k})
PY
done
