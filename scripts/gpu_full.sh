# full round evidence: tests, smoke, default bench, rocprof kernel trace of the bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 400 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo exit=$? >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo exit=$? >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 400 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-600
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1; echo exit=$? >> gpurun_out/prof/run.log
tail -1 gpurun_out/prof/run.log; ls gpurun_out/prof | head
