cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# fp32 token-major GEMM, K-split shape (fgemm3_kernel) against the 2-wave shape (fgemm2_kernel)
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "large_batch or batch32 or long_sequence" > gpurun_out/t15.log 2>&1; echo exit=$? >> gpurun_out/t15.log; grep -a "fp32 large\|passed\|failed\|Error\|error" gpurun_out/t15.log | tail -8 | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
for v in 0 1 2 3; do
SAID_FGEMM_V2=$v timeout 300 $L > gpurun_out/s_v$v.log 2>&1
python - <<PY
import json
s=open('gpurun_out/s_v$v.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('SAID_FGEMM_V2=$v', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
done
SAID_TG_DBG=1 timeout 300 $L > gpurun_out/s_noepi.log 2>&1
python - <<PY
import json
s=open('gpurun_out/s_noepi.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
r=d['roofline']
print('no epilogue', d['value'], 'step', r['unet_step']['ms_loop_per_step'], {k:(round(v['us']/v['launches'],1),v['launches']) for k,v in r['by_kernel'].items() if 'tgemm' in k or 'prep' in k})
PY
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_f32.txt 2>&1; grep -A78 "one denoise step" gpurun_out/prof_summary_b32_f32.txt | grep "fgemm\|denoise" | cut -c1-150
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
