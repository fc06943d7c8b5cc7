cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "bf16" 2>&1 | grep -E "bf16|passed|failed" | head
for B in 1 32; do timeout 300 python bench.py --batch $B --num_steps 50 --steps 2 --warmup 1 --dtype bf16 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('bf16 B=$B', d['value'], d['ms_per_step'])"; done
