# round 5: HIP runtime knobs that could touch the per-node cost of the step graph (kernarg placement, graph packet capture, fence scope)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/env.txt
run() { echo "== $*" | tee -a gpurun_out/r5/env.txt; env "$@" timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-190 | tee -a gpurun_out/r5/env.txt; }
for rep in 1 2; do
  run X=0
  run HIP_FORCE_DEV_KERNARG=1
  run HIP_FORCE_DEV_KERNARG=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run AMD_OPT_FLUSH=0
  run ROC_USE_FGS_KERNARG=0
done
echo "== node_floor default / dev kernarg" | tee -a gpurun_out/r5/env.txt
timeout 100 scripts/ubench/node_floor 2>&1 | head -2 | tee -a gpurun_out/r5/env.txt
HIP_FORCE_DEV_KERNARG=1 timeout 100 scripts/ubench/node_floor 2>&1 | head -2 | tee -a gpurun_out/r5/env.txt
HIP_FORCE_DEV_KERNARG=0 timeout 100 scripts/ubench/node_floor 2>&1 | head -2 | tee -a gpurun_out/r5/env.txt
