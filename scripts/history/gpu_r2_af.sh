cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# steps per captured graph at B=1 (default 10)
for n in 10 25 50 100; do
SAID_SPG=$n timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline > gpurun_out/af.log 2>&1; echo "SAID_SPG=$n $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/af.log | tr '\n' ' ')"
done
