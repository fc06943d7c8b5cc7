# round 5: stchain on bf16 operands (bf16 mode, large batches): bf16 tests, cfg2 A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/bfchain.txt
echo "== bf16 tests" | tee -a gpurun_out/r5/bfchain.txt
timeout 1500 python -m pytest tests -m gpu -q -k "bf16 or out_sched_tm or ensemble or token_major or batch32" 2>&1 | grep -v amdgpu.ids | tail -30 | tee -a gpurun_out/r5/bfchain.txt
for v in 1 0 1 0; do
  echo "== cfg2 st_chain_bf16=$v" | tee -a gpurun_out/r5/bfchain.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option st_chain_bf16=$v 2>&1 | tail -1 | cut -c1-330 | tee -a gpurun_out/r5/bfchain.txt
done
