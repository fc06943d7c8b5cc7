cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x > gpurun_out/t1.log 2>&1; echo exit=$? >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline > gpurun_out/bench2.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench2.log') if l.startswith('{')][-1]); print('B=1', d['value'], d['ms_per_step']); r=d['roofline']; print(r['unet_step'])
for k,v in r['by_kernel'].items(): print(k, v)"
