cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16" > gpurun_out/t8a.log 2>&1; echo exit=$? >> gpurun_out/t8a.log; tail -3 gpurun_out/t8a.log | cut -c1-300
grep -E "bf16" gpurun_out/t8a.log | cut -c1-200
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/g_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/g_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16"
for bb in 2 4; do
SAID_UNET_TGEMM_MIN=0 timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --batch $bb --num_steps 50 --dtype bf16 > gpurun_out/g_b${bb}_tg.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/g_b${bb}_tg.log | tr '\n' ' '; echo " <- B=$bb bf16 tgemm forced"
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --batch $bb --num_steps 50 --dtype bf16 > gpurun_out/g_b${bb}.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/g_b${bb}.log | tr '\n' ' '; echo " <- B=$bb bf16 default"
done
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_bf16.txt 2>&1; head -16 gpurun_out/prof_summary_b32_bf16.txt
