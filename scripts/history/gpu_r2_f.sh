# full parity suite + the three bench lines on the current build
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/f_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/f_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16"
timeout 300 $L > gpurun_out/f_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/f_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32"
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/t7.log 2>&1; echo exit=$? >> gpurun_out/t7.log; tail -4 gpurun_out/t7.log | cut -c1-300
grep -E "FAILED|Error|bf16" gpurun_out/t7.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo exit=$? >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-1200
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_bf16.txt 2>&1; head -16 gpurun_out/prof_summary_b32_bf16.txt
