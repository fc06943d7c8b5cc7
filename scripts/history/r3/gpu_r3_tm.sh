# round 3: the token-major activation path (xgemm_kernel) — focused parity tests + large-batch bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s -k "large_batch or ragged or batch32 or edge_shapes or batch_driver_64 or bf16_loop_cfg" > gpurun_out/r3_tm_tests.log 2>&1
echo "exit=$?" >> gpurun_out/r3_tm_tests.log
tail -40 gpurun_out/r3_tm_tests.log
for dt in bf16 f32; do
  timeout 300 python bench.py --batch 32 --num_steps 50 --dtype $dt --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/r3_tm_bench_$dt.log 2>&1
  python - <<PY
import json
ln=[l for l in open('gpurun_out/r3_tm_bench_$dt.log') if l.startswith('{')]
if ln:
    d=json.loads(ln[-1]); r=d['roofline']
    print('$dt', d['value'], d['ms_per_step'], r['unet_step']['ms_loop_per_step'], r['unet_step']['launches'])
    for k,v in r['by_kernel'].items(): print('   ', k, v)
else:
    print(open('gpurun_out/r3_tm_bench_$dt.log').read()[-2000:])
PY
done
