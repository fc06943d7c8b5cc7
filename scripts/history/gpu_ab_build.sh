# A/B of two BUILDS on one box: scripts/gpu_ab_build.sh "-DFOO=1" "-DFOO=2" [extra bench args]
# (rebuilds gemm_lds.hip with SAID_EXTRA_DEFS each time; 2 alternations x 2 bench runs)
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for i in 1 2; do
  for defs in "$A" "$B"; do
    touch said_amd/csrc/gemm_lds.hip
    SAID_EXTRA_DEFS="$defs" python -m said_amd.build > /dev/null 2>&1 || { echo "build failed for $defs"; exit 1; }
    for j in 1 2; do
      timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('[$defs]', d['ms_per_step'])"
    done
  done
done
