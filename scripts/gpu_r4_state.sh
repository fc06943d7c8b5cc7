# round 4: state of the tree on the box — the -m gpu suite (durations), smoke(), the default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1500 python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r4/suite.log 2>&1; echo "suite exit=$?" | tee -a gpurun_out/r4/suite.log
tail -25 gpurun_out/r4/suite.log | cut -c1-160
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 gpurun_out/r4/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/r4/bench_default.log 2>&1; echo "bench exit=$?"
tail -c 5000 gpurun_out/r4/bench_default.log
