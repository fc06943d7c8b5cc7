# A/B within one box: alternate two environments, 3 bench runs each, print ms per pass
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"
for i in 1 2 3; do
  for cfg in "$A" "$B"; do
    env $cfg timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$cfg', d['ms_per_step'])"
  done
done
