# where does tgemm's time go: epilogue / K loop knock-out experiment (results are wrong with SAID_TG_DBG != 0; timing only)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --batch 32 --num_steps 50 --dtype bf16"
for d in 0 1 2 3; do
SAID_TG_DBG=$d timeout 300 $L > gpurun_out/j_dbg$d.log 2>&1; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/j_dbg$d.log | tr '\n' ' '; echo " <- 256-row tiles, SAID_TG_DBG=$d"
done
for d in 0 1; do
SAID_NO_TGEMM256=1 SAID_TG_DBG=$d timeout 300 $L > gpurun_out/j_old_dbg$d.log 2>&1; grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/j_old_dbg$d.log | tr '\n' ' '; echo " <- 128-row tiles (UNet tgemm needs 256: audio only), SAID_TG_DBG=$d"
done
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
SAID_TG_DBG=1 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_dbg1.txt 2>&1; head -8 gpurun_out/prof_summary_dbg1.txt
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
SAID_TG_DBG=2 timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_dbg2.txt 2>&1; head -8 gpurun_out/prof_summary_dbg2.txt
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
