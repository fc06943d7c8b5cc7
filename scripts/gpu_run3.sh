cd $GRAFT_REPO_ROOT
for i in 1 2; do
for arch in gfx950 "gfx950:xnack-"; do
  SAID_OFFLOAD_ARCH="$arch" python -m said_amd.build --force > /dev/null 2>&1 || { echo "build failed $arch"; continue; }
  for j in 1 2; do timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('[$arch]', d['ms_per_step'])"; done
done; done
