# kconv_body (K-long up-path convolutions as straight-line blocks): parity, headline A/B against the block loop, shader-clock stamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t15
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x -k "kconv" > gpurun_out/r6t15/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|kconv vs|Error" gpurun_out/r6t15/tests.log | tail -12
for v in 0 1 1; do
  echo "== headline kconv=$v" | tee -a gpurun_out/r6t15/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t15/ab.txt
done
for v in 0 1; do
  echo "== cfg4 (30 s edit, 100 steps) kconv=$v" | tee -a gpurun_out/r6t15/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t15/ab.txt
done
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r6t15/clk_build.log 2>&1; echo "stamp build exit=$?"
CLK_DETAIL=1,13,14 timeout 300 python scripts/debug_clocks.py 2 600 > gpurun_out/r6t15/phase_clocks_b1.txt 2>&1; echo "clocks exit=$?"
grep -A9 "launch 13\|launch 14" gpurun_out/r6t15/phase_clocks_b1.txt
