cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# bf16 B=32: knock-outs of the token-major GEMMs (1 = no epilogue, 2 = no K loop, 3 = neither)
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
for d in 0 1 2 3; do
SAID_TG_DBG=$d timeout 300 $L > gpurun_out/ab2.log 2>&1
python - <<PY
import json
s=open('gpurun_out/ab2.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('bf16 SAID_TG_DBG=$d step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
done
./scripts/ubench/mfma_rate > gpurun_out/ubench_mfma_rate.txt 2>&1; cat gpurun_out/ubench_mfma_rate.txt
