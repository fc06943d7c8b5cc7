cd $GRAFT_REPO_ROOT
SAID_NO_UGEMM=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "not bf16" 2>&1 | tail -2
SAID_NO_MT=1 SAID_NO_FUSE_SCHED=1 SAID_NO_CONV_IN=1 SAID_SPG=1 timeout 900 python -m pytest tests/test_gpu_parity.py -q -x 2>&1 | tail -2
