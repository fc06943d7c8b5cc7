# audio encoder bf16 projections on tgemm256_kernel<256> (8 waves, 256 x 256 tile) against tgemm_kernel<128, SB>: dev-knobs build, 32 clips, encoder only matters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t23
for v in sb big; do
  opt=""; envs=""
  [ $v = big ] && { opt="--debug_option tgemm_sb=0"; envs="SAID_TGEMM_BALANCE=0"; }
  env $envs timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r6t23/tr -o $v -- python bench.py --batch 32 --num_steps 10 --steps 1 --warmup 1 --dtype bf16 --no_cpu_baseline --no_roofline --no_secondary --ab_lib said_amd/lib/ab_devknobs.so $opt > gpurun_out/r6t23/run_$v.log 2>&1
  python scripts/prof_summary.py $(find gpurun_out/r6t23/tr -name "${v}_results.db" | head -1) > gpurun_out/r6t23/trace_$v.txt 2>&1
  echo "== $v"; grep "tgemm" gpurun_out/r6t23/trace_$v.txt | head -5
done
find gpurun_out/r6t23/tr -name "*.db" -delete
