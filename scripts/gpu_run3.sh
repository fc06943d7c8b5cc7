cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
for B in 2 4 32; do
timeout 600 python bench.py --batch $B --steps 1 --warmup 1 --num_steps 50 --no_cpu_baseline > gpurun_out/bench_b$B.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_b$B.log') if l.startswith('{')][-1]); print('B=$B', d['value'], d['ms_per_step'], d['roofline']['unet_step']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)" || tail -5 gpurun_out/bench_b$B.log
done
B=32
SAID_QKV_UGEMM=1 timeout 600 python bench.py --batch $B --steps 1 --warmup 1 --num_steps 50 --no_cpu_baseline > gpurun_out/bench_b$B.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_b$B.log') if l.startswith('{')][-1]); print('B=$B qkv ugemm', d['value'], d['ms_per_step'], d['roofline']['unet_step']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)" || tail -5 gpurun_out/bench_b$B.log
