# round 4: bf16 large-batch parity tests + configs[2] with said_debug_option values (one process per run, alternating)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -q -x -s -k "bf16 or tm_acts or ragged or clip_groups or opt_in or ensemble" > gpurun_out/r4/cfg2_tests.log 2>&1; echo "tests exit=$?"
grep -E "passed|failed|error|Error" gpurun_out/r4/cfg2_tests.log | tail -3
for rep in 1 2 3; do for opt in ${@:-nop=0}; do
timeout 300 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --debug_option $opt 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 $opt', d['value'], d['ms_per_step'], d['config'].get('graph_nodes_per_step'))"
done; done | tee gpurun_out/r4/cfg2_opts.txt
