# round 4: split-fp16 attention products (attn.hip PM == 2) — accuracy tests, then same-box A/B of said_debug_option attn_split=0/1 on
# the headline, configs[4] (editing, 30 s) and configs[3] per GPU (B = 32 fp32, 100 of its 1000 steps)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -q -x -s -k "split_fp16" > gpurun_out/r4/attn_split_tests.log 2>&1; echo "tests exit=$?"
grep -E "attn_split\]|passed|failed|rror" gpurun_out/r4/attn_split_tests.log | tail -12
B="--no_cpu_baseline --no_roofline --no_secondary"
for rep in 1 2; do for opt in attn_split=0 attn_split=1; do
timeout 300 python bench.py --steps 3 --warmup 1 $B --debug_option $opt 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg1 $opt', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --edit --seconds 30 --num_steps 100 --steps 3 --warmup 1 $B --debug_option $opt 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg4 $opt', d['value'], d['ms_per_step'])"
timeout 300 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 $B --debug_option $opt 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3/100 $opt', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r4/attn_split_ab.txt
