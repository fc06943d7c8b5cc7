cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('   ms', d['ms_per_step'])"; done
