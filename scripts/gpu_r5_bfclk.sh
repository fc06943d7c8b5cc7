cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r5/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/debug_clocks_bf16.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/phase_clocks_stchain_bf16.txt; echo "exit=$?"
head -40 gpurun_out/r5/phase_clocks_stchain_bf16.txt
