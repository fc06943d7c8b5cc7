cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t5
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round6.py -m gpu -q -x > gpurun_out/r6t5/tests.log 2>&1; echo "tests exit=$?"; tail -3 gpurun_out/r6t5/tests.log
for rep in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t5/ab.txt
done
timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t5/ab.txt
timeout 600 python bench.py --batch 32 --steps 1 --warmup 1 --num_steps 100 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t5/ab.txt
