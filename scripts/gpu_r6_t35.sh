# K-long convolutions as NB = 1 kconv_body under CONCURRENT clip groups too (kconv = 2) against the default (not under concurrency): 4-8 clips
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t35
for b in 4 5 6 7 8; do for v in -1 2; do
  echo "== $b clips x 100 steps kconv=$v" | tee -a gpurun_out/r6t35/ab.txt
  timeout 600 python bench.py --batch $b --num_steps 100 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option kconv=$v 2>&1 | tail -1 | cut -c1-160 | tee -a gpurun_out/r6t35/ab.txt
done; done
