cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t7
for rep in 1 2; do
for v in 0 8; do
  echo "== cfg4 attn_ks=$v (pre-split K / V)" | tee -a gpurun_out/r6t7/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option attn_ks=$v 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t7/ab.txt
done; done
