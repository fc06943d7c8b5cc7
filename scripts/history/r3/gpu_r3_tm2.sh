cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -s -k "large_batch or ragged or batch32 or edge_shapes or batch_driver_64 or bf16_loop_cfg" > gpurun_out/r3_tm_tests.log 2>&1
echo "exit=$?" >> gpurun_out/r3_tm_tests.log
tail -12 gpurun_out/r3_tm_tests.log
for cfg in "bf16" "bf16 xgemm_ntw=1" "bf16 xgemm_ntw=2" "f32" "f32 xgemm_ntw=2"; do
  timeout 300 python scripts/profile_stages.py 32 600 $cfg 2>&1 | tail -45
done > gpurun_out/r3_tm_stages.log 2>&1
cat gpurun_out/r3_tm_stages.log
