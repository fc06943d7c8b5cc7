cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# sustained MFMA ceiling of the chip, then the knock-out experiment on the fp32 token-major GEMM
./scripts/ubench/mfma_peak > gpurun_out/ubench_mfma_peak.txt 2>&1; cat gpurun_out/ubench_mfma_peak.txt
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for d in 0 1 2 3; do
SAID_TG_DBG=$d timeout 300 $L > gpurun_out/n_dbg$d.log 2>&1
python - <<PY
import json
s=open('gpurun_out/n_dbg$d.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('SAID_TG_DBG=$d step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
done
find gpurun_out -name "*.db" -delete
