cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "ragged_length" > gpurun_out/t19.log 2>&1; echo exit=$? >> gpurun_out/t19.log; grep -a "sample\|single-clip\|passed\|failed\|Error\|error" gpurun_out/t19.log | tail -22 | cut -c1-200
