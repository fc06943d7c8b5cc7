# round 5: stchain everywhere it is eligible: suite, default bench with secondaries, cfg4 (edit, T = 1800) and 2 / 4 / 6-clip batches A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== suite" | tee gpurun_out/r5/chain5.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee -a gpurun_out/r5/chain5.txt
echo "== default bench" | tee -a gpurun_out/r5/chain5.txt
timeout 900 python bench.py 2>&1 | tail -1 > gpurun_out/r5/bench_default.json; cut -c1-300 gpurun_out/r5/bench_default.json | tee -a gpurun_out/r5/chain5.txt
for v in 1 0; do
  echo "== cfg4 edit 30 s st_chain=$v" | tee -a gpurun_out/r5/chain5.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/chain5.txt
  for b in 2 4 6; do
    echo "== batch $b x 100 steps st_chain=$v" | tee -a gpurun_out/r5/chain5.txt
    timeout 600 python bench.py --batch $b --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/chain5.txt
  done
done
