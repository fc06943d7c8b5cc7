cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2 3; do for lib in old new; do timeout 300 python scripts/ab_libs.py said_amd/lib/ab_$lib.so 32 50 2>&1 | grep "ms per step"; done; done
for lib in old new; do timeout 300 python scripts/ab_libs.py said_amd/lib/ab_$lib.so 1 400 fp32 2>&1 | grep "ms per step"; done
