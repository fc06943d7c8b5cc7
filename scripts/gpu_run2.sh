cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline > gpurun_out/bench2.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench2.log') if l.startswith('{')][-1]); print('B=1', d['value'], d['ms_per_step'], d['config']['graph_nodes_per_step']); r=d['roofline']; print(r['unet_step']); print(json.dumps(r['by_kernel'], indent=0))"
SAID_GEGLU_NB=1 timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/bench3.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench3.log') if l.startswith('{')][-1]); print('GEGLU_NB=1', d['value'], d['ms_per_step'])"
