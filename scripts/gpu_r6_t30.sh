# fp32 fused tail: GroupNorm coefficients from the q/k/v GEMM / its operand preparation for EVERY fp32 variant (one workgroup per tile too): parity + configs[3] / 5-8 clips A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t30
timeout 1200 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py tests/test_gpu_round3.py -m gpu -q -x > gpurun_out/r6t30/tests.log 2>&1; echo "tests exit=$?"; tail -2 gpurun_out/r6t30/tests.log
for v in 0 1 0 1; do
  echo "== cfg3 share (32 clips x 100 steps) chain_coef=$v" | tee -a gpurun_out/r6t30/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option chain_coef=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t30/ab.txt
done
for v in 0 1; do
  echo "== 8 clips x 100 steps chain_coef=$v" | tee -a gpurun_out/r6t30/ab.txt
  timeout 600 python bench.py --batch 8 --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option chain_coef=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t30/ab.txt
done
