# stchain: weight ring primed behind the attention tile's staging — the other configurations (ab_prev.so = the build before)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t34
for rep in 1 2; do for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== cfg2 (32 clips x 50 steps, bf16) $lib" | tee -a gpurun_out/r6t34/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t34/ab.txt
  echo "== cfg3 share (32 clips x 100 steps) $lib" | tee -a gpurun_out/r6t34/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t34/ab.txt
done; done
for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== cfg4 $lib" | tee -a gpurun_out/r6t34/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t34/ab.txt
done
timeout 600 python scripts/ab_equal.py said_amd/lib/ab_prev.so save /tmp/ref.pt 2>&1 | tail -1
timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so cmp /tmp/ref.pt 2>&1 | tail -1
