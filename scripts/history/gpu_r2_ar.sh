cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# bf16 64-row tile: one register set at five workgroups per CU (all 1216 workgroups of a 192-wide launch resident) vs two sets at four
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "bf16_unet_large_batch or ragged_length" > gpurun_out/t28.log 2>&1; echo exit=$? >> gpurun_out/t28.log; tail -2 gpurun_out/t28.log | cut -c1-200
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype bf16"
for rep in 1 2 3; do
for v in 0 1; do
SAID_BF_OCC5=$v timeout 300 $L > gpurun_out/ar.log 2>&1
echo "rep $rep SAID_BF_OCC5=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ar.log | tr '\n' ' ')"
done; done
