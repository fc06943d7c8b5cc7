cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# bf16 GEGLU tile: 128 VGPRs (5 spilled) at four workgroups per CU vs 133 at three
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype bf16"
for rep in 1 2 3; do
for v in 0 1; do
SAID_BF_GEGLU_OCC4=$v timeout 300 $L > gpurun_out/au.log 2>&1
echo "rep $rep SAID_BF_GEGLU_OCC4=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/au.log | tr '\n' ' ')"
done; done
