# round 5: attention K / V tiles two ahead (shipped) against one ahead (ab_pf1.so): tests, same-box A/B on the headline and configs[4], traces
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/attnpf.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_round4.py -m gpu -x -q 2>&1 | tail -2 | tee -a gpurun_out/r5/attnpf.txt
for rep in 1 2; do for v in shipped pf1; do
  if [ $v = shipped ]; then L=""; else L="--ab_lib said_amd/lib/ab_pf1.so"; fi
  echo "== headline $v" | tee -a gpurun_out/r5/attnpf.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $L 2>&1 | tail -1 | cut -c1-190 | tee -a gpurun_out/r5/attnpf.txt
done; done
for v in shipped pf1; do
  if [ $v = shipped ]; then L=""; else L="--ab_lib said_amd/lib/ab_pf1.so"; fi
  echo "== cfg4 $v" | tee -a gpurun_out/r5/attnpf.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $L 2>&1 | tail -1 | cut -c1-190 | tee -a gpurun_out/r5/attnpf.txt
done
for v in shipped pf1; do
  if [ $v = shipped ]; then L=""; else L="--ab_lib said_amd/lib/ab_pf1.so"; fi
  rm -rf gpurun_out/r5/tr_q
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 $L > gpurun_out/r5/run_q.log 2>&1
  echo "== trace $v" | tee -a gpurun_out/r5/attnpf.txt
  python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "b1_results.db" | head -1) 2>&1 | grep "attn_kernel<1\|one denoise" | head -2 | cut -c1-150 | tee -a gpurun_out/r5/attnpf.txt
done
rm -rf gpurun_out/r5/tr_q
