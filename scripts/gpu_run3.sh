cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "bf16" 2>&1 | grep -E "bf16|passed|failed|Error|assert" | head -20
