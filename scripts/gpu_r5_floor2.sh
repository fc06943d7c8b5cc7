# round 5: node floor micro-benchmarks + the engine's floor build without a profiler attached
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/floor2.txt
true
true
true
true
for ns in 200 1000; do
  echo "== engine floor build, no profiler, num_steps $ns" | tee -a gpurun_out/r5/floor2.txt
  timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps $ns --ab_lib said_amd/lib/ab_floor.so 2>&1 | tail -1 | cut -c1-260 | tee -a gpurun_out/r5/floor2.txt
done
