# round 4: in-situ rocprofv3 kernel traces of the four bench configurations + the three tests that failed on the first state run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -q -s -k "ragged_length_matches or opt_in_vs_oracle or two_streams_equal or bf16" > gpurun_out/r4/bf16_tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|bf16|tm_acts|clip groups" gpurun_out/r4/bf16_tests.log | cut -c1-220 | tail -60
trace() {  # name, bench flags
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r4/tr -o $name -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/r4/run_$name.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/r4/tr -name "${name}_results.db" | head -1) > gpurun_out/r4/trace_$name.txt 2>&1
  echo "trace $name: $(head -2 gpurun_out/r4/trace_$name.txt | tail -1)"
}
trace b1 --num_steps 200
trace cfg2_b32_bf16 --batch 32 --num_steps 50 --dtype bf16
trace cfg3_b32_f32 --batch 32 --num_steps 50
trace cfg4_edit --seconds 30 --num_steps 100 --edit
find gpurun_out/r4/tr -name "*.db" -delete
