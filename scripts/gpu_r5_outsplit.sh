# round 5: out_sched_kernel on split-fp16 products + preloaded header: parity subset, A/B on the headline and cfg4, trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/outsplit.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -4 | tee -a gpurun_out/r5/outsplit.txt
for rep in 1 2; do for v in 1 0; do
  echo "== headline out_split=$v" | tee -a gpurun_out/r5/outsplit.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option out_split=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/outsplit.txt
done; done
rm -rf gpurun_out/r5/tr_os
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_os -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_os.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_os -name "b1_results.db" | head -1) > gpurun_out/r5/trace_b1_outsplit.txt 2>&1
rm -rf gpurun_out/r5/tr_os
grep "out_sched\|one denoise" gpurun_out/r5/trace_b1_outsplit.txt | cut -c1-150 | tee -a gpurun_out/r5/outsplit.txt
