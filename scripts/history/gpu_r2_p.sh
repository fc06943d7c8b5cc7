cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_f32.txt 2>&1; sed -n 1,14p gpurun_out/prof_summary_b32_f32.txt; grep -A90 "one denoise step" gpurun_out/prof_summary_b32_f32.txt | cut -c1-150
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
