# round 5: localise the split-fp16 GEMM deviations
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/race
RACE_FULL=30 timeout 900 python scripts/race_localise.py 3 22 600 8 2>&1 | tee gpurun_out/race/localise_split1.txt
timeout 600 python scripts/poison_ws.py 2>&1 | tee gpurun_out/race/poison.txt
