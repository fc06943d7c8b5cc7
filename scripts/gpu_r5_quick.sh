# round 5: quick check of a kernel change: stchain / guided-step tests, headline bench twice, headline trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/quick.txt
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3 | tee -a gpurun_out/r5/quick.txt
for rep in 1 2; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/quick.txt
done
rm -rf gpurun_out/r5/tr_q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_q.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "b1_results.db" | head -1) > gpurun_out/r5/trace_b1_quick.txt 2>&1
rm -rf gpurun_out/r5/tr_q
sed -n 1,14p gpurun_out/r5/trace_b1_quick.txt | cut -c1-150 | tee -a gpurun_out/r5/quick.txt
grep "one denoise" gpurun_out/r5/trace_b1_quick.txt | tee -a gpurun_out/r5/quick.txt
