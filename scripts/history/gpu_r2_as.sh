cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "fp32_unet_large_batch or ragged_length" > gpurun_out/t29.log 2>&1; echo exit=$? >> gpurun_out/t29.log; tail -2 gpurun_out/t29.log | cut -c1-200
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50"
for rep in 1 2 3; do
for v in 0 1; do
SAID_F32_OCC5=$v timeout 300 $L > gpurun_out/as.log 2>&1
echo "rep $rep SAID_F32_OCC5=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/as.log | tr '\n' ' ')"
done; done
