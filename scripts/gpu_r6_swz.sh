cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash scripts/gpu_ab.sh EQ=1 cfg2 base lib:said_amd/lib/ab_prev.so
timeout 600 python -m pytest tests -m gpu -q -x -k "audio or direct_to_lds or encoder" 2>&1 | tail -3
mkdir -p gpurun_out/swz
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/swz -o b -- python bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary --batch 32 --num_steps 10 --dtype bf16 > gpurun_out/swz/run.log 2>&1
python scripts/pmc_generic_summary.py $(find gpurun_out/swz -name "b_results.db" | head -1) tgemm256d > gpurun_out/swz/conf.txt; cat gpurun_out/swz/conf.txt
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/swz/tr -o t -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --batch 32 --num_steps 50 --dtype bf16 > gpurun_out/swz/run_t.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/swz/tr -name "t_results.db" | head -1) | head -8
find gpurun_out/swz -name "*.db" -delete
