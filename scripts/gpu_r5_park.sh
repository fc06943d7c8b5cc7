# round 5: fp32 fused tail, K / V window parked in front of (shipped) or behind (ab_parklate.so) the first barrier: same-box A/B with traces
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/park.txt
for rep in 1 2; do for v in shipped late; do
  if [ $v = shipped ]; then L=""; else L="--ab_lib said_amd/lib/ab_parklate.so"; fi
  echo "== $v" | tee -a gpurun_out/r5/park.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $L 2>&1 | tail -1 | cut -c1-190 | tee -a gpurun_out/r5/park.txt
done; done
for v in shipped late; do
  if [ $v = shipped ]; then L=""; else L="--ab_lib said_amd/lib/ab_parklate.so"; fi
  rm -rf gpurun_out/r5/tr_q
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 $L > gpurun_out/r5/run_q.log 2>&1
  echo "== trace $v" | tee -a gpurun_out/r5/park.txt
  python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "b1_results.db" | head -1) 2>&1 | grep "stchain_kernel<false>\|one denoise" | head -2 | cut -c1-150 | tee -a gpurun_out/r5/park.txt
done
rm -rf gpurun_out/r5/tr_q
