# LayerNorm statistics of the LayerNorm'ed GEMMs: a lane merges ONE token's partials: bit-identity against the previous build (ab_prev.so) + A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t27
timeout 600 python scripts/ab_equal.py said_amd/lib/ab_prev.so save /tmp/ref.pt 2>&1 | tail -1
timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so cmp /tmp/ref.pt 2>&1 | tail -13 | tee gpurun_out/r6t27/equal.txt
for rep in 1 2; do
for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== headline $lib" | tee -a gpurun_out/r6t27/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t27/ab.txt
done; done
for lib in "--ab_lib said_amd/lib/ab_prev.so" ""; do
  echo "== strict fp32 headline $lib" | tee -a gpurun_out/r6t27/ab.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --dtype f32_strict --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t27/ab.txt
done
