cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SAID_DEV=1
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s -k "large_batch or ragged or batch32 or edge_shapes or batch_driver_64 or bf16_loop_cfg" > gpurun_out/r3_tm_tests.log 2>&1
echo "exit=$?" >> gpurun_out/r3_tm_tests.log
tail -5 gpurun_out/r3_tm_tests.log
for cfg in "bf16" "bf16 xgemm_dbg=1" "bf16 xgemm_ntw=1" "f32" "f32 tm_acts=0"; do
  timeout 300 python scripts/profile_stages.py 32 600 $cfg 2>&1 | tail -65
done > gpurun_out/r3_tm_stages2.log 2>&1
grep "^sum" gpurun_out/r3_tm_stages2.log
