# round 3 evidence batch: full -m gpu suite, smoke(), the default bench line (headline + secondary configurations), kernel traces of
# the four configurations, SQ counters of the headline, and the large-batch schedules' A/B (token-major activations forced off / on: the default is on in bf16 mode, off in fp32 mode)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests -m gpu -q --durations=15 -s > gpurun_out/final/suite.log 2>&1; echo "suite exit=$?" | tee -a gpurun_out/final/suite.log
tail -22 gpurun_out/final/suite.log | cut -c1-150
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke exit=$?"; tail -1 gpurun_out/final/smoke.log
timeout 900 python bench.py --steps 5 --warmup 2 > gpurun_out/final/bench_default.log 2>&1; echo "bench exit=$?"
trace() {  # name, bench flags
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/final/tr -o $name -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/final/run_$name.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/final/tr -name "${name}_results.db" | head -1) > gpurun_out/final/trace_$name.txt 2>&1
  echo "trace $name: $(head -1 gpurun_out/final/trace_$name.txt)"
}
trace b1 --num_steps 200
trace cfg2_b32_bf16 --batch 32 --num_steps 50 --dtype bf16
trace cfg3_b32_f32 --batch 32 --num_steps 50
trace cfg4_edit --seconds 30 --num_steps 100 --edit
trace cfg2_b32_bf16_hybrid --batch 32 --num_steps 50 --dtype bf16 --tm_acts 0
find gpurun_out/final/tr -name "*.db" -delete
rm -rf gpurun_out/final/sq; mkdir -p gpurun_out/final/sq
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/final/sq -o sq -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/final/sq/run.log 2>&1; echo "sq exit=$?"
python scripts/pmc_generic_summary.py $(find gpurun_out/final/sq -name "sq*_results.db" | head -1) ugemm attn_kernel out_sched conv_in > gpurun_out/final/sq_b1.txt 2>&1
find gpurun_out/final/sq -name "*.db" -delete
for dt in bf16 f32; do for v in "--tm_acts 0" "--tm_acts 1"; do
  timeout 300 python bench.py --batch 32 --num_steps 50 --dtype $dt --steps 3 --warmup 1 --no_cpu_baseline --no_roofline $v 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt $v', d['value'], d['ms_per_step'], d['config']['graph_nodes_per_step'])"
done; done | tee gpurun_out/final/ab_large_batch.txt
du -sh gpurun_out/final
