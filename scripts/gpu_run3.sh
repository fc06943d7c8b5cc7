cd $GRAFT_REPO_ROOT
for gs in 1.0 2.0; do timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --guidance_scale $gs > gpurun_out/bench2.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench2.log') if l.startswith('{')][-1]); print('gs=$gs', d['value'], d['ms_per_step']); r=d['roofline']
for k,v in r['by_kernel'].items(): print('  ', k, v)"; done
