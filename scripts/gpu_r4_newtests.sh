# round 4: the new / re-bounded tests (-s: their measured values)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_gpu_round4.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -s -k "round4 or ensemble or wide or ragged_shapes or bf16 or batch32 or tm_acts or opt_in or front_end" > gpurun_out/r4/newtests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|bf16|wide|FAILED|Error|assert" gpurun_out/r4/newtests.log | cut -c1-330 | tail -50
