cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# GEGLU on a 128-row K-split tile (8 waves) vs the 64-row tile, both precisions, in situ
SAID_GEGLU_WR4=3 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "large_batch or ragged_length" > gpurun_out/t31.log 2>&1; echo exit=$? >> gpurun_out/t31.log; tail -2 gpurun_out/t31.log | cut -c1-200
for dt in f32 bf16; do
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype $dt"
for rep in 1 2 3; do
for v in 0 3; do
SAID_GEGLU_WR4=$v timeout 300 $L > gpurun_out/av.log 2>&1
echo "$dt rep $rep SAID_GEGLU_WR4=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/av.log | tr '\n' ' ')"
done; done; done
