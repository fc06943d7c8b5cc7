# round 6: the three-slice fused tail — its tests, the round-5 chain tests, then headline A/B (slices 1 vs 3) twice
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t2
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x -k "three_slice or trained_like" > gpurun_out/r6t2/slices.log 2>&1; echo "slices exit=$?"
grep -E "passed|failed|three slices|trained-like" gpurun_out/r6t2/slices.log | tail -14
timeout 900 python -m pytest tests/test_gpu_round5.py -m gpu -q -x > gpurun_out/r6t2/round5.log 2>&1; echo "round5 exit=$?"; tail -2 gpurun_out/r6t2/round5.log
for rep in 1 2; do
for v in 1 3; do
  echo "== headline st_chain_slices=$v" | tee -a gpurun_out/r6t2/ab.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_slices=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t2/ab.txt
done; done
for v in 1 3; do
  echo "== 2 clips x 100 steps st_chain_slices=$v" | tee -a gpurun_out/r6t2/ab.txt
  timeout 600 python bench.py --batch 2 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_slices=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t2/ab.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t2/tr -o cfg1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r6t2/run_trace.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r6t2/tr -name "cfg1_results.db" | head -1) > gpurun_out/r6t2/trace_cfg1.txt 2>&1
find gpurun_out/r6t2/tr -name "*.db" -delete
head -16 gpurun_out/r6t2/trace_cfg1.txt
for v in 0 -4 8; do
  echo "== cfg4 attn_ks=$v" | tee -a gpurun_out/r6t2/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_ks=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t2/ab.txt
done
