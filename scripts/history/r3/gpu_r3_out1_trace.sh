#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1; do
  rm -rf gpurun_out/o1tr
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/o1tr -o t -- python bench.py --batch 32 --num_steps 20 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --clip_groups 1 --debug_option f32_out1_tm=$v > gpurun_out/o1tr_run$v.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/o1tr -name "t_results.db" | head -1) > gpurun_out/r3_out1_trace$v.txt 2>&1
  find gpurun_out/o1tr -name "*.db" -delete
done
echo done
