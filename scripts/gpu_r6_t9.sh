cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t9
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q -s -x > gpurun_out/r6t9/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|slice" gpurun_out/r6t9/tests.log | tail -14
for v in 1 -1; do
  echo "== cfg4 st_chain_slices=$v" | tee -a gpurun_out/r6t9/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_slices=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t9/ab.txt
  echo "== 3 clips x 100 steps st_chain_slices=$v" | tee -a gpurun_out/r6t9/ab.txt
  timeout 600 python bench.py --batch 3 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_slices=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t9/ab.txt
done
for v in 2 3; do
  echo "== headline st_chain_slices=$v" | tee -a gpurun_out/r6t9/ab.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_slices=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t9/ab.txt
done
