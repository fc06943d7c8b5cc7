cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# fp32 token-major GEMM (fgemm_kernel): parity of the large-batch fp32 path, then the B=32 fp32 step with and without it
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "large_batch or batch32 or long_sequence or golden_and_oracle" > gpurun_out/t13.log 2>&1; echo exit=$? >> gpurun_out/t13.log; grep -a "max err\|passed\|failed\|Error\|error" gpurun_out/t13.log | tail -12 | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L > gpurun_out/m_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/m_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32 fgemm<2,3>"
SAID_FGEMM_WM4=1 timeout 300 $L > gpurun_out/m_b32_f32_wm4.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/m_b32_f32_wm4.log | tr '\n' ' '; echo " <- B=32 f32 fgemm<4,3>"
SAID_NO_UNET_FGEMM=1 timeout 300 $L > gpurun_out/m_b32_f32_no.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/m_b32_f32_no.log | tr '\n' ' '; echo " <- B=32 f32 channel-major (round-1 path)"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_f32.txt 2>&1; head -16 gpurun_out/prof_summary_b32_f32.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
