cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "long_sequence or 30s or batch32" 2>&1 | tail -15
