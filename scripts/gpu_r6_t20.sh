# attn_kernel with vector-register-form MFMAs (launch bounds: two waves per SIMD): tests + bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t20
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_round4.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r6t20/tests.log 2>&1; echo "tests exit=$?"; tail -2 gpurun_out/r6t20/tests.log
for rep in 1 2; do
  echo "== headline" | tee -a gpurun_out/r6t20/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t20/ab.txt
done
for v in 0 1; do
  echo "== cfg4 (30 s edit, 100 steps) attn_2q=$v" | tee -a gpurun_out/r6t20/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_2q=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t20/ab.txt
done
echo "== cfg3 share (32 clips x 50 steps)" | tee -a gpurun_out/r6t20/ab.txt
timeout 600 python bench.py --batch 32 --num_steps 50 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t20/ab.txt
echo "== 3 clips x 100 steps" | tee -a gpurun_out/r6t20/ab.txt
timeout 600 python bench.py --batch 3 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t20/ab.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t20/tr -o cfg1 -- python bench.py --num_steps 200 --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/r6t20/run_trace.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r6t20/tr -name "cfg1_results.db" | head -1) > gpurun_out/r6t20/trace_cfg1.txt 2>&1
find gpurun_out/r6t20/tr -name "*.db" -delete
sed -n 2,14p gpurun_out/r6t20/trace_cfg1.txt
