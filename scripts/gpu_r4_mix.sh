# round 4: (1) self-attention query tiles per workgroup at configs[4]'s size, (2) sc1 stores in the round-2/3 large-batch kernels (fp32, configs[3]), (3) the tests touched
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 600 python -m pytest tests -m gpu -q -s -k "identical_clips or long_sequence or editing_30s or edit or wide" > gpurun_out/r4/mix_tests.log 2>&1; echo "tests exit=$?"; grep -E "passed|failed|identical|T=1800" gpurun_out/r4/mix_tests.log | cut -c1-200 | tail -6
for rep in 1 2; do for qw in 0 2 3 4 -1; do
  timeout 300 python bench.py --seconds 30 --num_steps 100 --edit --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --debug_option attn_qw=$qw 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 attn_qw=$qw', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r4/attn_qw_cfg4.txt
VARIANT_SRC=tgemm_dev.h AB_B=32 AB_N=50 AB_DT=fp32 bash scripts/gpu_r4_ab_defs.sh sc1:-DSAID_TG_ST_SC1
