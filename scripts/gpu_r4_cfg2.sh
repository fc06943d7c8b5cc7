# round 4: the bf16 large-batch path after a kernel change — its parity tests, the configs[2] bench line, then the shader-clock stamps
# (stamp build made on the box, which is discarded afterwards)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or tm_acts or ragged or clip_groups or opt_in" > gpurun_out/r4/cfg2_tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|error" gpurun_out/r4/cfg2_tests.log | tail -3
for i in 1 2; do
timeout 300 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2', d['value'], d['ms_per_step'], d['config'].get('graph_nodes_per_step'))"
done
SAID_ALLOW_SCRATCH=1 SAID_EXTRA_DEFS=-DSAID_CLK_STAMPS python -m said_amd.build --force > gpurun_out/r4/clk_build.log 2>&1; echo "build exit=$?"
timeout 300 python scripts/rgemm_clocks.py 32 600 > gpurun_out/r4/rgemm_clocks.txt 2>&1; echo "clocks exit=$?"
grep "^launch" gpurun_out/r4/rgemm_clocks.txt | head -48 | cut -c1-250
