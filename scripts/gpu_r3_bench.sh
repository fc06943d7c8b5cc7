# round 3: the default bench line (headline + secondary configurations)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r3_bench_default.log 2>&1
echo "exit=$?" >> gpurun_out/r3_bench_default.log
tail -c 6000 gpurun_out/r3_bench_default.log
