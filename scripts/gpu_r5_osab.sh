# round 5: where do out_sched_kernel's 13 us go?  Variants: 1 returns at entry, 2 no epilogue operands / step chain, 3 no GroupNorm, 4 no MFMA loop
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/osab.txt
for v in base os1 os2 os3 os4; do
  if [ $v = base ]; then L=""; else L="--ab_lib said_amd/lib/ab_$v.so"; fi
  rm -rf gpurun_out/r5/tr_os
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_os -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 $L > gpurun_out/r5/run_os.log 2>&1
  echo "== $v" | tee -a gpurun_out/r5/osab.txt
  python scripts/prof_summary.py $(find gpurun_out/r5/tr_os -name "b1_results.db" | head -1) 2>&1 | grep "out_sched\|one denoise step\|conv_in_kernel" | cut -c1-150 | tee -a gpurun_out/r5/osab.txt
done
rm -rf gpurun_out/r5/tr_os
