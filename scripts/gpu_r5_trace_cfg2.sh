cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr -o cfg2 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --batch 32 --num_steps 50 --dtype bf16 > gpurun_out/r5/run_cfg2.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr -name "cfg2_results.db" | head -1) > gpurun_out/r5/trace_cfg2_b32_bf16.txt 2>&1
find gpurun_out/r5/tr -name "*.db" -delete
head -70 gpurun_out/r5/trace_cfg2_b32_bf16.txt
