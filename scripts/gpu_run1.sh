cd $GRAFT_REPO_ROOT
timeout 200 python bench.py --steps 2 --warmup 1 --num_steps 100 --no_cpu_baseline --no_roofline > gpurun_out/bench_a.log 2>&1; echo exit=$? >> gpurun_out/bench_a.log
tail -3 gpurun_out/bench_a.log
timeout 300 python bench.py --steps 1 --warmup 1 --num_steps 100 --no_cpu_baseline > gpurun_out/bench_b.log 2>&1; echo exit=$? >> gpurun_out/bench_b.log
tail -3 gpurun_out/bench_b.log
