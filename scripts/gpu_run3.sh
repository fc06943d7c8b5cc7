cd $GRAFT_REPO_ROOT
SAID_NO_UGEMM=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "not bf16" 2>&1 | tail -4
