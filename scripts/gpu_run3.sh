cd $GRAFT_REPO_ROOT
for cfg in SAID_A=1 SAID_BF16_NB1=1 "SAID_BF16_NB1=1 SAID_QKV_NB=1 SAID_GEGLU_NB=1"; do
B=32
env $cfg timeout 600 python bench.py --batch $B --steps 1 --warmup 1 --num_steps 50 --dtype bf16 --no_cpu_baseline > gpurun_out/bench_b$B.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench_b$B.log') if l.startswith('{')][-1]); print('$cfg B=$B', d['value'], d['ms_per_step'], d['roofline']['unet_step']['ms_graph_replay']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)" || tail -5 gpurun_out/bench_b$B.log
done
