cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L="python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype bf16"
for rep in 1 2 3; do
for v in default 3 1 2; do
if [ $v = default ]; then unset SAID_TGEMM_SMALL; else export SAID_TGEMM_SMALL=$v; fi
timeout 300 $L > gpurun_out/ak.log 2>&1
echo "rep $rep SAID_TGEMM_SMALL=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ak.log | tr '\n' ' ')"
done; done
