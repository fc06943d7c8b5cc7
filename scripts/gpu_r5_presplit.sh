# round 5: k / v stored pre-split by the q/k/v GEMM (attn_kernel<PM = 3>): tests, A/B on the headline and configs[4], trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/presplit.txt
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py -m gpu -x -q -s -k "presplit or stchain or guided or unet" 2>&1 | grep -v amdgpu.ids | grep "pre-split\|passed\|failed\|Error" | tee -a gpurun_out/r5/presplit.txt
for rep in 1 2; do for v in 1 0; do
  echo "== headline attn_presplit=$v" | tee -a gpurun_out/r5/presplit.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_presplit=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/presplit.txt
done; done
for v in 1 0; do
  echo "== cfg4 attn_presplit=$v" | tee -a gpurun_out/r5/presplit.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_presplit=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/presplit.txt
done
rm -rf gpurun_out/r5/tr_q
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_q -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 > gpurun_out/r5/run_q.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_q -name "b1_results.db" | head -1) 2>&1 | grep "attn_kernel\|ugemm_kernel<3\|ugemm_kernel<2\|one denoise" | head -8 | cut -c1-150 | tee -a gpurun_out/r5/presplit.txt
rm -rf gpurun_out/r5/tr_q
