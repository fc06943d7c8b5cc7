cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# GroupNorm partials in tile-major layout: parity subset, then the B=1 headline step and the B=32 steps
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py -q -m gpu -s -x -k "not 1000 and not 997 and not editing" > gpurun_out/t12.log 2>&1; echo exit=$? >> gpurun_out/t12.log; tail -3 gpurun_out/t12.log | cut -c1-300
timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline > gpurun_out/l_b1.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|sum_kernel_us": [0-9.]*' gpurun_out/l_b1.log | tr '\n' ' '; echo " <- B=1 headline"
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/l_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/l_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16"
timeout 300 $L > gpurun_out/l_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/l_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32"
find gpurun_out -name "*.db" -delete
