# round 2: parity suite (with the measured errors printed), smoke, default bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/t1.log 2>&1; echo exit=$? >> gpurun_out/t1.log; tail -5 gpurun_out/t1.log
grep -E "max abs err|max err|bf16|g9 |teacher|vae|FAILED|Error" gpurun_out/t1.log | head -60
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo exit=$? >> gpurun_out/smoke.log; tail -2 gpurun_out/smoke.log
timeout 500 python bench.py > gpurun_out/bench_default.log 2>&1; echo exit=$? >> gpurun_out/bench_default.log; tail -2 gpurun_out/bench_default.log | cut -c1-1500
