# round 5: steps per captured graph (10 today) on the headline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -f gpurun_out/r5/spg.txt
for rep in 1 2; do for n in 10 25 50 100; do
  echo "== steps_per_graph=$n" | tee -a gpurun_out/r5/spg.txt
  timeout 600 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --debug_option steps_per_graph=$n 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r5/spg.txt
done; done
