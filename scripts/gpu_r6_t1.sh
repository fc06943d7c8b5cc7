# round 6, first contact: the new precision-guard tests, then the whole -m gpu suite, then a short default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t1
timeout 1200 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x > gpurun_out/r6t1/round6.log 2>&1; echo "round6 exit=$?"
grep -E "passed|failed|Error|error" gpurun_out/r6t1/round6.log | tail -5
timeout 1500 python -m pytest tests -m gpu -q --durations=10 --deselect tests/test_gpu_round6.py > gpurun_out/r6t1/suite.log 2>&1; echo "suite exit=$?"
tail -3 gpurun_out/r6t1/suite.log
timeout 900 python bench.py --steps 3 --warmup 1 > gpurun_out/r6t1/bench.log 2>&1; echo "bench exit=$?"
tail -1 gpurun_out/r6t1/bench.log | cut -c1-600
