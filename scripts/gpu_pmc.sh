# HBM traffic counters (separate passes, no tracing domains besides kernel-trace), per MI355X_MICROARCH.md §HBM
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc; mkdir -p gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc -o pmc_$c -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline > gpurun_out/pmc/run_$c.log 2>&1; echo "$c exit=$?" >> gpurun_out/pmc/run_$c.log; tail -1 gpurun_out/pmc/run_$c.log
done
ls -la gpurun_out/pmc | head
