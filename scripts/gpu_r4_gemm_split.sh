# round 4: fgemm_kernel on split-fp16 operands — accuracy / determinism tests, then same-box A/B of said_debug_option gemm_split=0/1 on configs[3]'s per-GPU share
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests -m gpu -q -x -s -k "split_fp16 or capacity or clip_groups or large_batch or b32 or 32_clips or batched_driver" > gpurun_out/r4/gemm_split_tests.log 2>&1; echo "tests exit=$?"
grep -E "attn_split\]|passed|failed|rror" gpurun_out/r4/gemm_split_tests.log | tail -12
B="--no_cpu_baseline --no_roofline --no_secondary"
for rep in 1 2; do for opt in gemm_split=0 gemm_split=1; do
timeout 300 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 $B --debug_option $opt 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3/100 $opt', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r4/gemm_split_ab.txt
