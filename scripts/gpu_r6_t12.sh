# per-wave shader-clock stamps of the K-long up-block convolutions (launches 13 / 14 / 18 / 19 of the forward schedule) beside a plain one (1)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t12
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r6t12/clk_build.log 2>&1; echo "stamp build exit=$?"
CLK_DETAIL=1,2,13,14,18,19,23 timeout 300 python scripts/debug_clocks.py 2 600 > gpurun_out/r6t12/phase_clocks_b1.txt 2>&1; echo "clocks exit=$?"
grep -A9 "launch 13\|launch 14" gpurun_out/r6t12/phase_clocks_b1.txt
