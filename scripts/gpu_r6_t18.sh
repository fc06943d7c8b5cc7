# full -m gpu suite on the kconv sources + the rounds rule at 4 clips / configs[4]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t18
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r6t18/tests.log 2>&1; echo "tests exit=$?"; tail -3 gpurun_out/r6t18/tests.log
for v in 0 4096; do
  echo "== cfg4 (30 s edit, 100 steps) kconv_max_tiles=$v" | tee -a gpurun_out/r6t18/ab.txt
  timeout 600 python bench.py --seconds 30 --num_steps 100 --edit --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv_max_tiles=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t18/ab.txt
  for b in 2 4 5; do
  echo "== $b clips x 100 steps kconv_max_tiles=$v" | tee -a gpurun_out/r6t18/ab.txt
  timeout 600 python bench.py --batch $b --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv_max_tiles=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t18/ab.txt
  done
done
