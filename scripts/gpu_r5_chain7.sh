# round 5: stchain as the fp32 default at every batch size: suite, determinism soak with three clip groups, default bench (+ secondaries), in-situ traces
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== suite" | tee gpurun_out/r5/chain7.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee -a gpurun_out/r5/chain7.txt
for i in 1 2 3 4; do
  echo "== soak #$i: 32 clips, three clip groups" | tee -a gpurun_out/r5/chain7.txt
  DET_GEMM_SPLIT=1 timeout 300 python scripts/attn_split_det.py 1 32 3 2>&1 | grep attn_split | cut -c1-200 | tee -a gpurun_out/r5/chain7.txt
done
echo "== default bench" | tee -a gpurun_out/r5/chain7.txt
timeout 1200 python bench.py 2>&1 | tail -1 > gpurun_out/r5/bench_default.json; cut -c1-400 gpurun_out/r5/bench_default.json | tee -a gpurun_out/r5/chain7.txt
trace() {  # name, bench flags
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr -o $name -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/r5/run_$name.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/r5/tr -name "${name}_results.db" | head -1) > gpurun_out/r5/trace_$name.txt 2>&1
  echo "trace $name: $(sed -n 1p gpurun_out/r5/trace_$name.txt)"
}
trace b1 --num_steps 200
trace cfg3_b32_f32 --batch 32 --num_steps 50
trace cfg4_edit --seconds 30 --num_steps 100 --edit
find gpurun_out/r5/tr -name "*.db" -delete
