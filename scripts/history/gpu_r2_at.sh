cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# preparation kernel: 20 GroupNorm-partial loads up front (one round trip at T = 600, 96 VGPRs with 5 spilled) vs 10 (two round trips, 84 VGPRs)
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "large_batch or ragged_length" > gpurun_out/t30.log 2>&1; echo exit=$? >> gpurun_out/t30.log; tail -2 gpurun_out/t30.log | cut -c1-200
for dt in bf16 f32; do
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype $dt"
for rep in 1 2 3; do
for v in 10 20; do
SAID_PREP_NL=$v timeout 300 $L > gpurun_out/at.log 2>&1
echo "$dt rep $rep SAID_PREP_NL=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/at.log | tr '\n' ' ')"
done; done; done
