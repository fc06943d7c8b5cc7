# stchain_kernel reads the GroupNorm coefficients the q/k/v GEMM finalised (chain_coef): parity + headline / configs[4] A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t24
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q -s -x -k "chain or sliced or stchain or tail" > gpurun_out/r6t24/tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|coefficients from|Error" gpurun_out/r6t24/tests.log | tail -10
for v in 0 1 0 1; do
  echo "== headline chain_coef=$v" | tee -a gpurun_out/r6t24/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline --debug_option chain_coef=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t24/ab.txt
done
for v in 0 1; do
  echo "== cfg4 chain_coef=$v" | tee -a gpurun_out/r6t24/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option chain_coef=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t24/ab.txt
done
