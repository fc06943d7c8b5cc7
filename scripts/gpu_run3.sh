cd $GRAFT_REPO_ROOT
for cfg in SAID_A=1; do
for dt in f32 bf16; do
for B in 8 32; do
env $cfg timeout 600 python bench.py --batch $B --steps 1 --warmup 1 --num_steps 50 --dtype $dt --no_cpu_baseline > gpurun_out/bench_b$B.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_b$B.log') if l.startswith('{')][-1]); print('$cfg $dt B=$B', d['value'], d['ms_per_step'], d['roofline']['unet_step']['ms_graph_replay']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)" || tail -5 gpurun_out/bench_b$B.log
done; done; done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -3
