cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t11
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round4.py tests/test_gpu_round3.py -m gpu -q -s -x > gpurun_out/r6t11/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|pre-split K|Error" gpurun_out/r6t11/tests.log | tail -8
for v in 0 1; do
  echo "== cfg3 (32 clips x 100 steps) attn_presplit=$v" | tee -a gpurun_out/r6t11/ab.txt
  timeout 900 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_presplit=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t11/ab.txt
  echo "== cfg3 one group attn_presplit=$v" | tee -a gpurun_out/r6t11/ab.txt
  timeout 900 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --clip_groups 1 --debug_option attn_presplit=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t11/ab.txt
done
echo "== 12 clips x 100 steps" | tee -a gpurun_out/r6t11/ab.txt
for v in 0 1; do timeout 900 python bench.py --batch 12 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_presplit=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t11/ab.txt; done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t11/tr -o cfg3 -- python bench.py --batch 32 --num_steps 50 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/r6t11/run_trace.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r6t11/tr -name "cfg3_results.db" | head -1) > gpurun_out/r6t11/trace_cfg3.txt 2>&1
find gpurun_out/r6t11/tr -name "*.db" -delete
head -12 gpurun_out/r6t11/trace_cfg3.txt
