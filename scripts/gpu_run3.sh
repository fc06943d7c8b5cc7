cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout 300 python scripts/bf16_probe.py 2>&1 | grep "fp32 err"
for B in 1 32; do timeout 300 python bench.py --batch $B --num_steps 50 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('B=$B', d['value'], d['ms_per_step'])"; done
