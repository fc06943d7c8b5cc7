cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# the fused token-local chain kernel (xattn.hip) at LARGE batch, fp32, in situ
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for rep in 1 2; do
for v in off on; do
unset SAID_XATTN SAID_XATTN_MAX_WGS
[ $v = on ] && export SAID_XATTN=1 SAID_XATTN_MAX_WGS=100000000
timeout 300 $L > gpurun_out/al.log 2>&1
python - <<PY
import json
s=open('gpurun_out/al.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('rep $rep xattn $v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], r['unet_step']['launches'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'xattn' in k or 'band' in k or 'NB1,KS8,store' in k or 'NB2,KS8,store' in k})
PY
done; done
