# round 5: stchain at larger batches: 8 / 12 / 16 / 32 clips fp32, chain (max tiles lifted, large on) vs the shipped schedule; clip groups 1 / 3 at 32
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/chain6.txt
run() { echo "== $*" | tee -a gpurun_out/r5/chain6.txt; timeout 900 python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline "$@" 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/chain6.txt; }
for b in 8 12 16; do
  run --batch $b --num_steps 100
  run --batch $b --num_steps 100 --debug_option st_chain_max_tiles=100000 --debug_option st_chain_large=1
done
run --batch 32 --num_steps 100
run --batch 32 --num_steps 100 --debug_option st_chain_max_tiles=100000 --debug_option st_chain_large=1
run --batch 32 --num_steps 100 --debug_option st_chain_max_tiles=100000 --debug_option st_chain_large=1 --clip_groups 1
run --batch 32 --num_steps 100 --debug_option st_chain_max_tiles=100000 --debug_option st_chain_large=1 --clip_groups 2
