# stchain: the weight ring primed behind the attention tile staging (-DSAID_CHAIN_RING_LATE) against the shipped order
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t33
timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so save /tmp/ref.pt 2>&1 | tail -1
timeout 600 python scripts/ab_equal.py said_amd/lib/ab_ringlate.so cmp /tmp/ref.pt 2>&1 | tail -1 | tee gpurun_out/r6t33/equal.txt
for rep in 1 2; do for lib in "" "--ab_lib said_amd/lib/ab_ringlate.so"; do
  echo "== headline $lib" | tee -a gpurun_out/r6t33/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t33/ab.txt
done; done
