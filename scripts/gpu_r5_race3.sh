cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/race
SAID_AB_LIB=said_amd/lib/ab_o2.so RACE_SPLIT=0 RACE_ATTN=1 RACE_FULL=0 timeout 900 python scripts/race_localise.py 3 22 600 10 7 10 2>&1 | cut -c1-600 | tee gpurun_out/race/localise_o2_pattern.txt
