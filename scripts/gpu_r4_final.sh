# round 4 evidence batch: the -m gpu suite with durations, smoke(), the default bench line, the driver's command line, in-situ kernel traces of the four
# configurations, SQ counters of the bf16 large-batch step, shader-clock stamps of the persistent kernels (stamp build made last: the box is discarded),
# then the PMC traffic passes (scripts/gpu_r3_traffic.sh) on the same snapshot.    bash scripts/gpu_r4_final.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/final; mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -s > gpurun_out/final/suite.log 2>&1; echo "suite exit=$?" | tee -a gpurun_out/final/suite.log
grep -E "passed|failed" gpurun_out/final/suite.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 gpurun_out/final/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/final/bench_default.log 2>&1; echo "bench exit=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_like.log 2>&1; echo "bench (driver's command line) exit=$?"
trace() {  # name, bench flags
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/final/tr -o $name -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/final/run_$name.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/final/tr -name "${name}_results.db" | head -1) > gpurun_out/final/trace_$name.txt 2>&1
  echo "trace $name: $(sed -n 1p gpurun_out/final/trace_$name.txt)"
}
trace b1 --num_steps 200
trace cfg2_b32_bf16 --batch 32 --num_steps 50 --dtype bf16
trace cfg3_b32_f32 --batch 32 --num_steps 50
trace cfg4_edit --seconds 30 --num_steps 100 --edit
find gpurun_out/final/tr -name "*.db" -delete
rm -rf gpurun_out/final/sq; mkdir -p gpurun_out/final/sq
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/final/sq -o sq -- python bench.py --batch 32 --num_steps 20 --dtype bf16 --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/final/sq/run.log 2>&1; echo "sq exit=$?"
python scripts/pmc_generic_summary.py $(find gpurun_out/final/sq -name "sq*_results.db" | head -1) rgemm battn out_sched conv_in tgemm > gpurun_out/final/sq_cfg2.txt 2>&1
find gpurun_out/final/sq -name "*.db" -delete
if [ -z "$FINAL_SKIP_TRAFFIC" ]; then bash scripts/gpu_r3_traffic.sh > gpurun_out/final/traffic_run.log 2>&1; tail -3 gpurun_out/final/traffic_run.log; fi   # (FINAL_SKIP_TRAFFIC=1: the traffic passes ran first, on the same sources, so that the bench lines above carry a fresh stamp)
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/final/clk_build.log 2>&1; echo "stamp build exit=$?"
timeout 300 python scripts/rgemm_clocks.py 32 600 > gpurun_out/final/rgemm_clocks.txt 2>&1; echo "clocks exit=$?"
du -sh gpurun_out/final
