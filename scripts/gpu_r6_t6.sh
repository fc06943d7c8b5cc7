cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t6
for g in 1 2 3; do
  echo "== cfg2 bf16 clip_groups=$g" | tee -a gpurun_out/r6t6/ab.txt
  timeout 600 python bench.py --batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline --clip_groups $g 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t6/ab.txt
done
for b in 27 54; do
  echo "== bf16 batch=$b (1026 / 2052 fused-tail workgroups: whole rounds of 512)" | tee -a gpurun_out/r6t6/ab.txt
  timeout 600 python bench.py --batch $b --num_steps 50 --dtype bf16 --steps 3 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline 2>&1 | tail -1 | cut -c1-200 | tee -a gpurun_out/r6t6/ab.txt
done
