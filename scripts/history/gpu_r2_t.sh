cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# knock-outs inside the K-split fp32 GEMM's loop; counter list for the memory path
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for v in 0 1 2 4 7; do
SAID_FG_KO=$v SAID_TG_DBG=1 timeout 300 $L > gpurun_out/t_ko$v.log 2>&1
python - <<PY
import json
s=open('gpurun_out/t_ko$v.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('fgemm3 KO=$v (no epilogue)', 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k})
PY
done
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC|TA|TD|SQ|SQC)_[A-Z0-9_]+" | sort -u > gpurun_out/counter_names.txt; wc -l gpurun_out/counter_names.txt
