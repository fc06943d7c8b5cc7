# round 5: split-fp16 ugemm, staging v2 (8-byte writes, permuted k order): suite, headline A/B, trace, shader-clock stamps (stamp build last)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== suite" | tee gpurun_out/r5/sp3.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee -a gpurun_out/r5/sp3.txt
for v in 1 0 1; do
  echo "== bench ugemm_split=$v" | tee -a gpurun_out/r5/sp3.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --debug_option ugemm_split=$v 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/r5/sp3.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_b1.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr -name "b1_results.db" | head -1) > gpurun_out/r5/trace_b1_sp3.txt 2>&1
find gpurun_out/r5/tr -name "*.db" -delete
head -80 gpurun_out/r5/trace_b1_sp3.txt
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r5/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/debug_clocks.py 2 600 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/phase_clocks_b1_sp3.txt; echo "clocks exit=$?"
grep "^launch" gpurun_out/r5/phase_clocks_b1_sp3.txt | cut -c1-220
