# round 3: the whole -m gpu suite with per-test durations (output -> gpurun_out/r3_suite.log)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x --durations=40 -s > gpurun_out/r3_suite.log 2>&1
echo "exit=$?" >> gpurun_out/r3_suite.log
tail -60 gpurun_out/r3_suite.log
