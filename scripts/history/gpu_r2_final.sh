# round 2 evidence: parity suite, smoke, bench lines of configs[1], [2] (bf16 + fp32), [4]; kernel traces; PMC traffic of the dominant B=1 kernel
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/t_final.log 2>&1; echo exit=$? >> gpurun_out/t_final.log; tail -4 gpurun_out/t_final.log | cut -c1-300
grep -E "FAILED|Error" gpurun_out/t_final.log | head
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo exit=$? >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-400
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/bench_cfg3_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/bench_cfg3_bf16.log | tr '\n' ' '; echo " <- cfg3 bf16"
timeout 300 $L > gpurun_out/bench_cfg3_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/bench_cfg3_f32.log | tr '\n' ' '; echo " <- cfg3 f32"
timeout 400 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --seconds 30 --num_steps 100 --edit > gpurun_out/bench_cfg5.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_cfg5.log | tr '\n' ' '; echo " <- cfg5 editing 30 s / 100 steps"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b1.txt 2>&1; head -16 gpurun_out/prof_summary_b1.txt
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_bf16.txt 2>&1; head -12 gpurun_out/prof_summary_b32_bf16.txt
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_f32.txt 2>&1; head -12 gpurun_out/prof_summary_b32_f32.txt
rm -rf gpurun_out/prof gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc -o pmc_$c -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline > gpurun_out/pmc/run_$c.log 2>&1; echo "$c exit=$?"
done
python scripts/pmc_summary.py $(find gpurun_out/pmc -name "pmc_FETCH_SIZE*_results.db" | head -1) $(find gpurun_out/pmc -name "pmc_WRITE_SIZE*_results.db" | head -1) gpurun_out/traffic_latest.json > gpurun_out/pmc_summary_b1.txt 2>&1; head -14 gpurun_out/pmc_summary_b1.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof gpurun_out/pmc
