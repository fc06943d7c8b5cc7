# ugemm_body: weights requested behind every other request of the phase (-DSAID_UGEMM_W_LAST, said_amd/lib/ab_wlast.so) against the shipped order
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t16
for rep in 1 2; do
for lib in "" "--ab_lib said_amd/lib/ab_wlast.so"; do
  echo "== headline $lib" | tee -a gpurun_out/r6t16/ab.txt
  timeout 600 python bench.py --steps 5 --warmup 2 --no_cpu_baseline --no_secondary --no_roofline $lib 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t16/ab.txt
done; done

