# round 5: stchain launches confined to 1 / 2 / 4 / 8 XCDs (headline: 38 workgroups per launch; cfg4: 114)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/xcds.txt
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k "stchain or guided" 2>&1 | tail -2 | tee -a gpurun_out/r5/xcds.txt
for rep in 1 2; do for nx in 8 0 4 2; do
  echo "== headline st_chain_xcds=$nx" | tee -a gpurun_out/r5/xcds.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_xcds=$nx 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/xcds.txt
done; done
for nx in 8 0 4; do
  echo "== cfg4 st_chain_xcds=$nx" | tee -a gpurun_out/r5/xcds.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_xcds=$nx 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/xcds.txt
done
