# the default bench line and the driver's command line, on sources whose traces / traffic are already committed under profiles/ (in-situ times and PMC traffic current)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final
timeout 1500 python bench.py --steps 5 --warmup 2 > gpurun_out/final/bench_default.log 2>&1; echo "bench exit=$?"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver_like.log 2>&1; echo "bench (driver's command line) exit=$?"
tail -1 gpurun_out/final/bench_driver_like.log | cut -c1-300
timeout 300 python bench.py --rccl_at_one --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --no_roofline > gpurun_out/final/bench_rccl_one_rank.log 2>&1; echo "one-rank RCCL exit=$?"; tail -1 gpurun_out/final/bench_rccl_one_rank.log | cut -c1-300
