cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t4
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s > gpurun_out/r6t4/round6.log 2>&1; echo "round6 exit=$?"
grep -E "passed|failed|trained-like|three slices|split vs strict" gpurun_out/r6t4/round6.log | tail -16
timeout 1500 python -m pytest tests -m gpu -q --durations=8 --deselect tests/test_gpu_round6.py > gpurun_out/r6t4/suite.log 2>&1; echo "suite exit=$?"
tail -4 gpurun_out/r6t4/suite.log
timeout 900 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary > gpurun_out/r6t4/bench.log 2>&1; echo "bench exit=$?"
tail -1 gpurun_out/r6t4/bench.log | cut -c1-300
