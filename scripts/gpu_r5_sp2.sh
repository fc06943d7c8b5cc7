# round 5: where the headline step's time is now (split-fp16 ugemm): in-situ kernel trace, then shader-clock stamps of every launch (stamp build on the box)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_b1.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr -name "b1_results.db" | head -1) > gpurun_out/r5/trace_b1_sp.txt 2>&1
find gpurun_out/r5/tr -name "*.db" -delete
head -40 gpurun_out/r5/trace_b1_sp.txt
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r5/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/debug_clocks.py 2 600 2>&1 | grep -v amdgpu.ids > gpurun_out/r5/phase_clocks_b1_sp.txt; echo "clocks exit=$?"
cat gpurun_out/r5/phase_clocks_b1_sp.txt | cut -c1-220
