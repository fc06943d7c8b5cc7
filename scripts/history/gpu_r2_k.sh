cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16 or batch32 or large_batch or audio_encoder" > gpurun_out/t11a.log 2>&1; echo exit=$? >> gpurun_out/t11a.log; tail -3 gpurun_out/t11a.log | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/k_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/k_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16 (LDS-staged K/V attention)"
timeout 300 $L > gpurun_out/k_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/k_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_bf16.txt 2>&1; head -12 gpurun_out/prof_summary_b32_bf16.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
