# full round evidence: tests, smoke, default bench, rocprof kernel trace of the bench command, PMC traffic passes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 500 python -m pytest tests -q -m gpu > gpurun_out/t1.log 2>&1; echo exit=$? >> gpurun_out/t1.log; tail -3 gpurun_out/t1.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo exit=$? >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
rm -rf gpurun_out/prof gpurun_out/pmc; mkdir -p gpurun_out/prof gpurun_out/pmc
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1; echo exit=$? >> gpurun_out/prof/run.log
tail -1 gpurun_out/prof/run.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc -o pmc_$c -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline > gpurun_out/pmc/run_$c.log 2>&1; echo "$c exit=$?" >> gpurun_out/pmc/run_$c.log; tail -1 gpurun_out/pmc/run_$c.log
done
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary.txt 2>&1; head -12 gpurun_out/prof_summary.txt
python scripts/pmc_summary.py $(find gpurun_out/pmc -name "pmc_FETCH_SIZE*_results.db" | head -1) $(find gpurun_out/pmc -name "pmc_WRITE_SIZE*_results.db" | head -1) profiles/traffic_latest.json > gpurun_out/pmc_summary.txt 2>&1; cat gpurun_out/pmc_summary.txt | head -12
cp profiles/traffic_latest.json gpurun_out/traffic_latest.json
timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-900
# phase clocks need the stamp sites compiled in (the product library has none); the box is discarded afterwards
touch said_amd/csrc/gemm_common.h; SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build > /dev/null 2>&1
timeout 200 python scripts/debug_clocks.py > gpurun_out/clk.log 2>&1; grep -c "^launch" gpurun_out/clk.log
