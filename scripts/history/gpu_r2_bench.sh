cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# the round's bench lines (committed under profiles/ as r02f_bench_*.json)
timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/bench_cfg3_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/bench_cfg3_bf16.log | tr '\n' ' '; echo " <- cfg3 bf16"
timeout 300 $L > gpurun_out/bench_cfg3_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/bench_cfg3_f32.log | tr '\n' ' '; echo " <- cfg3 f32"
timeout 400 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --seconds 30 --num_steps 100 --edit > gpurun_out/bench_cfg5.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/bench_cfg5.log | tr '\n' ' '; echo " <- cfg5 editing 30 s / 100 steps"
