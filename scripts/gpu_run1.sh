cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests -q -m gpu -x > gpurun_out/t1.log 2>&1; echo exit=$? >> gpurun_out/t1.log; tail -4 gpurun_out/t1.log
timeout 300 python bench.py --steps 2 --warmup 1 > gpurun_out/bench1.log 2>&1; echo exit=$? >> gpurun_out/bench1.log
tail -3 gpurun_out/bench1.log
