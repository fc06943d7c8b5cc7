cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# audio encoder (bf16): 128 x 128 tiles where the 256-row grid fills its rounds / rows badly; margin sweep (percent)
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
for rep in 1 2; do
for v in 15 30 5 0; do
SAID_TGEMM_BALANCE=$v timeout 300 $L > gpurun_out/aq.log 2>&1
echo "rep $rep SAID_TGEMM_BALANCE=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/aq.log | tr '\n' ' ')"
done; done
