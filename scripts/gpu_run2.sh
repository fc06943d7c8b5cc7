cd $GRAFT_REPO_ROOT
timeout 400 python -m pytest tests/test_gpu_cli.py -q -x > gpurun_out/t2.log 2>&1; echo exit=$? >> gpurun_out/t2.log; tail -25 gpurun_out/t2.log
