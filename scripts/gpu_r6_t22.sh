# prep_kernel knock-outs (configs[3] share, 32 clips x 20 steps, in-situ traces): 1 no in-kernel GroupNorm finalisation, 2 no SiLU / LayerNorm / pack, 3 no stores, 4 no loads
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t22
for k in 0 1 2 3 4; do
  lib=""; [ $k -gt 0 ] && lib="--ab_lib said_amd/lib/ab_prepko$k.so"
  timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t22/tr -o ko$k -- python bench.py --batch 32 --num_steps 20 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --clip_groups 1 $lib > gpurun_out/r6t22/run_$k.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/r6t22/tr -name "ko${k}_results.db" | head -1) > gpurun_out/r6t22/trace_$k.txt 2>&1
  echo "== knock-out $k: $(grep prep_kernel gpurun_out/r6t22/trace_$k.txt | head -1)"
done
find gpurun_out/r6t22/tr -name "*.db" -delete
