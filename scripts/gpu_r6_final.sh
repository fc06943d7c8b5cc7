# round 6 evidence batch (ONE gpurun call): the -m gpu suite with durations, smoke(), PMC traffic passes (scripts/gpu_r3_traffic.sh), in-situ kernel traces of the
# four configurations — each trace file starts with "# source_hash=<bench.source_hash()>" so that bench.py only quotes an in-situ time taken on the sources it runs —,
# the default bench line, the driver's command line, shader-clock stamps (stamp build made last).    bash scripts/gpu_r6_final.sh ; then HERE: python scripts/traffic_merge.py r06
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
SH=$(python -c "import bench; print(bench.source_hash())")
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -s > gpurun_out/final/suite.log 2>&1; echo "suite exit=$?" | tee -a gpurun_out/final/suite.log
grep -E "passed|failed" gpurun_out/final/suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 gpurun_out/final/smoke.log
if [ -z "$FINAL_SKIP_TRAFFIC" ]; then bash scripts/gpu_r3_traffic.sh > gpurun_out/final/traffic_run.log 2>&1; tail -3 gpurun_out/final/traffic_run.log; fi
trace() {  # name, bench flags
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/final/tr -o $name -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/final/run_$name.log 2>&1
  { echo "# source_hash=$SH   rocprofv3 --kernel-trace --stats -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary $*"; \
    python scripts/prof_summary.py $(find gpurun_out/final/tr -name "${name}_results.db" | head -1); } > gpurun_out/final/trace_$name.txt 2>&1
  echo "trace $name: $(sed -n 2p gpurun_out/final/trace_$name.txt)"
}
trace cfg1 --num_steps 200
trace cfg1_strict --num_steps 200 --dtype f32_strict
trace cfg2_bf16 --batch 32 --num_steps 50 --dtype bf16
trace cfg3_per_gpu_f32 --batch 32 --num_steps 50
trace cfg4_edit --seconds 30 --num_steps 100 --edit
find gpurun_out/final/tr -name "*.db" -delete
# (the committed traces are what bench.py's in_situ figure reads: copy them where it looks BEFORE the bench lines are taken)
for c in cfg1 cfg1_strict cfg2_bf16 cfg3_per_gpu_f32 cfg4_edit; do cp gpurun_out/final/trace_$c.txt profiles/trace_latest_$c.txt; done
timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/final/bench_default.log 2>&1; echo "bench exit=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_like.log 2>&1; echo "bench (driver's command line) exit=$?"
tail -1 gpurun_out/final/bench_driver_like.log | cut -c1-300
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/final/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/debug_clocks.py 2 600 > gpurun_out/final/phase_clocks_b1.txt 2>&1; echo "clocks exit=$?"
du -sh gpurun_out/final
