cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cli.py -q -m gpu -x > gpurun_out/t26.log 2>&1; echo exit=$? >> gpurun_out/t26.log; tail -4 gpurun_out/t26.log | cut -c1-300
