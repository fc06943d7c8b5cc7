# attn2q_kernel: two / three query tiles per wave at T = 1800 (configs[4]) against one (attn_2q = 0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t21
timeout 900 python -m pytest tests/test_gpu_round6.py -m gpu -q -s -x -k "query_tiles_per_wave" > gpurun_out/r6t21/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|attn_2q=|Error" gpurun_out/r6t21/tests.log | tail -12
for v in 0 1 0 1; do
  echo "== cfg4 (30 s edit, 100 steps) attn_2q=$v" | tee -a gpurun_out/r6t21/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_2q=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t21/ab.txt
done
for v in 1; do
  echo "== headline attn_2q=$v (forced)" | tee -a gpurun_out/r6t21/ab.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_2q=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t21/ab.txt
  echo "== 3 clips x 100 steps attn_2q=$v (forced)" | tee -a gpurun_out/r6t21/ab.txt
  timeout 600 python bench.py --batch 3 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_2q=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t21/ab.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t21/tr -o cfg4 -- python bench.py --seconds 30 --num_steps 100 --edit --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/r6t21/run_trace.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r6t21/tr -name "cfg4_results.db" | head -1) > gpurun_out/r6t21/trace_cfg4.txt 2>&1
find gpurun_out/r6t21/tr -name "*.db" -delete
head -14 gpurun_out/r6t21/trace_cfg4.txt
