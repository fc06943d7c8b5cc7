# K-long convolutions as NB = 1 kconv_body workgroups where the chosen NB has no split-fp16 shape (configs[4], 3-8 clips): kconv_max_tiles 0 (off) against on
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t17
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round5.py -m gpu -q -x > gpurun_out/r6t17/tests.log 2>&1; echo "tests exit=$?"; tail -2 gpurun_out/r6t17/tests.log
for v in 0 100000; do
  echo "== cfg4 (30 s edit, 100 steps) kconv_max_tiles=$v" | tee -a gpurun_out/r6t17/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv_max_tiles=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t17/ab.txt
  for b in 3 4 6 8; do
  echo "== $b clips x 100 steps kconv_max_tiles=$v" | tee -a gpurun_out/r6t17/ab.txt
  timeout 600 python bench.py --batch $b --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv_max_tiles=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t17/ab.txt
  done
done
