cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/s1
timeout 1200 python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/s1/suite.log 2>&1; echo "suite exit=$?" | tee -a gpurun_out/s1/suite.log
tail -15 gpurun_out/s1/suite.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench_driver_like.log 2>&1; echo "bench exit=$?"
tail -1 gpurun_out/s1/bench_driver_like.log | cut -c1-400
