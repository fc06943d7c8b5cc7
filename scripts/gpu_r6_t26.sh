# lane_xor32 (v_permlane32_swap) instead of ds_bpermute for the cross-half exchanges: bit-identity against the previous build (ab_gnold.so) + A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t26
timeout 600 python scripts/ab_equal.py said_amd/lib/ab_gnold.so save /tmp/ref.pt 2>&1 | tail -1
timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so cmp /tmp/ref.pt 2>&1 | tail -14 | tee gpurun_out/r6t26/equal.txt
for rep in 1 2; do
for lib in "--ab_lib said_amd/lib/ab_gnold.so" ""; do
  echo "== headline $lib" | tee -a gpurun_out/r6t26/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t26/ab.txt
done; done
for lib in "--ab_lib said_amd/lib/ab_gnold.so" ""; do
  echo "== cfg4 $lib" | tee -a gpurun_out/r6t26/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t26/ab.txt
  echo "== cfg2 (32 clips x 50 steps, bf16) $lib" | tee -a gpurun_out/r6t26/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t26/ab.txt
done
