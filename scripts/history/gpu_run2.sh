cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline > gpurun_out/bench2.log 2>&1; python -c "
import json; d=json.loads([l for l in open('gpurun_out/bench2.log') if l.startswith('{')][-1]); print('B=1', d['value'], d['ms_per_step']); r=d['roofline']
for k,v in r['by_kernel'].items(): print(k, v)" || tail -5 gpurun_out/bench2.log; }
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py -q -x 2>&1 | tail -8
run SAID_X=1
timeout 200 python tests/debug_clocks.py > gpurun_out/clk.log 2>&1; grep -E "^launch +(1|2|3|5|6|7|8|9|10|23) " gpurun_out/clk.log
