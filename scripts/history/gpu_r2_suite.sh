cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/t_suite.log 2>&1; echo exit=$? >> gpurun_out/t_suite.log; tail -4 gpurun_out/t_suite.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
