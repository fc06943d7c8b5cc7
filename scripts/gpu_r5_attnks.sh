# round 5: self-attention key slices per workgroup at T = 600 (KS 4 -> 8) on the headline; also out_sched's interleaved GroupNorm chains (in the shipped build)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/attnks.txt
for rep in 1 2; do for ks in 0 8; do
  echo "== headline SAID_ATTN_KS=$ks (dev-knob build)" | tee -a gpurun_out/r5/attnks.txt
  if [ $ks = 0 ]; then unset SAID_ATTN_KS; else export SAID_ATTN_KS=$ks; fi
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --ab_lib said_amd/lib/ab_knobs.so 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/attnks.txt
done; done
export SAID_ATTN_KS=8
rm -rf gpurun_out/r5/tr_q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 --ab_lib said_amd/lib/ab_knobs.so > gpurun_out/r5/run_q.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "b1_results.db" | head -1) 2>&1 | grep "attn_kernel\|out_sched\|one denoise" | cut -c1-150 | tee -a gpurun_out/r5/attnks.txt
rm -rf gpurun_out/r5/tr_q
