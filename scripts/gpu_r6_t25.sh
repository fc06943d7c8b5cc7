# gn_finish with every lane summing its own group (no second LDS round trip): bit-identity against the previous build (ab_gnold.so) + headline A/B
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t25
timeout 600 python scripts/ab_equal.py said_amd/lib/ab_gnold.so save /tmp/ref.pt 2>&1 | tail -1
timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so cmp /tmp/ref.pt 2>&1 | tail -14 | tee gpurun_out/r6t25/equal.txt
for rep in 1 2; do
for lib in "--ab_lib said_amd/lib/ab_gnold.so" ""; do
  echo "== headline $lib" | tee -a gpurun_out/r6t25/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t25/ab.txt
done; done
