cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "large_batch or ragged_length or batch32 or engine_library" > gpurun_out/t25.log 2>&1; echo exit=$? >> gpurun_out/t25.log; tail -3 gpurun_out/t25.log | cut -c1-200
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16 --no_roofline > gpurun_out/am.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/am.log
