cd $GRAFT_REPO_ROOT
touch said_amd/csrc/gemm_common.h; SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build > /dev/null 2>&1
timeout 200 python tests/debug_clocks.py > gpurun_out/clk.log 2>&1; grep -c "^launch" gpurun_out/clk.log
