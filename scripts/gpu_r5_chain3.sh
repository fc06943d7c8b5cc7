# round 5: fused SpatialTransformer tail, self-contained roles: errors vs oracle, parity file, headline A/B, clock stamps
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 600 python scripts/chain_debug.py 2>&1 | grep -v amdgpu.ids | cut -c1-330 | tee gpurun_out/r5/chain_debug.txt
echo "== parity" | tee gpurun_out/r5/chain3.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -5 | tee -a gpurun_out/r5/chain3.txt
for v in 1 0 1; do
  echo "== bench st_chain=$v" | tee -a gpurun_out/r5/chain3.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain=$v 2>&1 | tail -1 | cut -c1-330 | tee -a gpurun_out/r5/chain3.txt
done
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r5/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/debug_clocks.py 2 600 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/phase_clocks_b1_chain.txt; echo "clocks exit=$?"
grep -A9 "^launch  5" gpurun_out/r5/phase_clocks_b1_chain.txt | cut -c1-200
