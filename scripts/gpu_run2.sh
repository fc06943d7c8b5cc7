cd $GRAFT_REPO_ROOT
SAID_MIN_LDS=98304 timeout 120 python tests/debug_clocks.py 2 600 > gpurun_out/clk_a.log 2>&1; grep "launch  1 \|launch  7 \|launch  9 " gpurun_out/clk_a.log
timeout 120 python tests/debug_clocks.py 2 600 > gpurun_out/clk_b.log 2>&1; grep "launch  1 \|launch  7 \|launch  9 " gpurun_out/clk_b.log
SAID_MIN_LDS=98304 timeout 200 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/bench2.log 2>&1; tail -1 gpurun_out/bench2.log | cut -c1-220
timeout 200 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/bench3.log 2>&1; tail -1 gpurun_out/bench3.log | cut -c1-220
