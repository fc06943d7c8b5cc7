cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -x -m gpu 2>&1 | tail -3
timeout 300 python bench.py --steps 3 --warmup 1 --no_cpu_baseline --no_roofline 2>&1 | tail -1 | cut -c1-330
