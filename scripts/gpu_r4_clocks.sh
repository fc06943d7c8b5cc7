# round 4: shader-clock stamps of the persistent kernels on a -DSAID_CLK_STAMPS build made on the box (the box is discarded afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r4/clk_build.log 2>&1; echo "build exit=$?"
timeout 300 python scripts/rgemm_clocks.py 32 600 > gpurun_out/r4/rgemm_clocks.txt 2>&1; echo "clocks exit=$?"
grep -c "^launch" gpurun_out/r4/rgemm_clocks.txt; head -60 gpurun_out/r4/rgemm_clocks.txt | cut -c1-260
