# round 5: bf16 stchain at two workgroups per CU (<= 80 KB LDS, 128 VGPRs): bf16 tests, cfg2 bench, trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/bf2.txt
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -x -q 2>&1 | tail -5 | tee -a gpurun_out/r5/bf2.txt
for rep in 1 2; do
  timeout 600 python bench.py --batch 32 --dtype bf16 --num_steps 50 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/bf2.txt
done
rm -rf gpurun_out/r5/tr_q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o c2 -- python bench.py --batch 32 --dtype bf16 --num_steps 50 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/r5/run_q.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "c2_results.db" | head -1) > gpurun_out/r5/trace_cfg2_bf2.txt 2>&1
rm -rf gpurun_out/r5/tr_q
sed -n 1,12p gpurun_out/r5/trace_cfg2_bf2.txt | cut -c1-150 | tee -a gpurun_out/r5/bf2.txt
grep "one denoise" gpurun_out/r5/trace_cfg2_bf2.txt | tee -a gpurun_out/r5/bf2.txt
