# round 5: fused SpatialTransformer tail (stchain.hip): suite, headline A/B (st_chain 1 / 0), trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== quick parity" | tee gpurun_out/r5/chain1.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -25 | tee -a gpurun_out/r5/chain1.txt
for v in 1 0 1; do
  echo "== bench st_chain=$v" | tee -a gpurun_out/r5/chain1.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain=$v 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/r5/chain1.txt
done
echo "== suite" | tee -a gpurun_out/r5/chain1.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee -a gpurun_out/r5/chain1.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_b1.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr -name "b1_results.db" | head -1) > gpurun_out/r5/trace_b1_chain1.txt 2>&1
find gpurun_out/r5/tr -name "*.db" -delete
head -60 gpurun_out/r5/trace_b1_chain1.txt
