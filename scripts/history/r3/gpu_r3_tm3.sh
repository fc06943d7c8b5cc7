cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp SAID_DEV=1
mkdir -p gpurun_out
for cfg in "bf16" "bf16 xgemm_dbg=1" "bf16 xgemm_dbg=4" "bf16 xgemm_dbg=8" "bf16 xgemm_dbg=2" "bf16 xgemm_dbg=15" "bf16 tm_acts=0"; do
  timeout 300 python scripts/profile_stages.py 32 600 $cfg 2>&1 | tail -45
done > gpurun_out/r3_tm_stages2.log 2>&1
grep "^sum" gpurun_out/r3_tm_stages2.log
