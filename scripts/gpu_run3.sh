cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "bf16" 2>&1 | grep -E "bf16|passed|failed|Error|assert" | head
timeout 300 python scripts/bf16_probe.py 2>&1 | tail -3
for B in 1 32; do
timeout 600 python bench.py --batch $B --steps 2 --warmup 1 --num_steps 50 --dtype bf16 --no_cpu_baseline > gpurun_out/bench_b$B.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_b$B.log') if l.startswith('{')][-1]); print('bf16 B=$B', d['value'], d['ms_per_step'], d['roofline']['unet_step']['ms_graph_replay']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)" || tail -5 gpurun_out/bench_b$B.log
done
