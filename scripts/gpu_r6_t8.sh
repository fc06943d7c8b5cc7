cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t8
run() { echo "== cfg3 (32 clips x 100 steps, fp32) $*" | tee -a gpurun_out/r6t8/ab.txt
  timeout 900 python bench.py --batch 32 --num_steps 100 --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline "$@" 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t8/ab.txt; }
run
run --clip_groups 1
run --clip_groups 1 --debug_option unet_tgemm_min_tokens=100000000
run --clip_groups 1 --debug_option unet_tgemm_min_tokens=100000000 --debug_option mt_wgs=100000000
run --debug_option unet_tgemm_min_tokens=100000000 --debug_option mt_wgs=100000000
