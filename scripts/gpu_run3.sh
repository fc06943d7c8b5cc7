cd $GRAFT_REPO_ROOT
for B in 8 16 32; do for w in 1024 512 256; do
SAID_MT_WGS=$w timeout 300 python bench.py --batch $B --num_steps 50 --steps 2 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('B=$B wgs/tile=$w', d['value'], d['ms_per_step'])"
done; done
