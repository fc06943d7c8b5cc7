# tgemm256d_kernel (direct-to-LDS 256 x 256 tile) in the bf16 audio encoder: parity + bit-identity + configs[2] A/B + trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t29
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_parity.py tests/test_gpu_round3.py -m gpu -q -s -x -k "direct or audio" > gpurun_out/r6t29/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|direct-to-LDS|Error|assert" gpurun_out/r6t29/tests.log | tail -10
for v in 0 1 0 1; do
  echo "== cfg2 (32 clips x 50 steps, bf16) tgemm_direct=$v" | tee -a gpurun_out/r6t29/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option tgemm_direct=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t29/ab.txt
done
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t29/tr -o cfg2 -- python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/r6t29/run_trace.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r6t29/tr -name "cfg2_results.db" | head -1) > gpurun_out/r6t29/trace_cfg2.txt 2>&1
find gpurun_out/r6t29/tr -name "*.db" -delete
sed -n 2,12p gpurun_out/r6t29/trace_cfg2.txt
