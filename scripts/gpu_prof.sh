cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 1 --warmup 1 --num_steps 100 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1; echo exit=$? >> gpurun_out/prof/run.log
tail -2 gpurun_out/prof/run.log
find gpurun_out/prof -type f | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo $f; head -30 $f
