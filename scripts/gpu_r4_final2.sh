# round 4, last evidence batch in ONE box call: PMC traffic passes first, merged on the box against the dominant-kernel names of a quick bench line, so that the
# bench lines of scripts/gpu_r4_final.sh (run next, FINAL_SKIP_TRAFFIC=1) carry a traffic stamp of exactly these sources
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/final0
bash scripts/gpu_r3_traffic.sh > gpurun_out/final0/traffic_run.log 2>&1; tail -1 gpurun_out/final0/traffic_run.log
timeout 600 python bench.py --steps 1 --warmup 1 --no_cpu_baseline > gpurun_out/final0/names.log 2>&1; echo "names bench exit=$?"
python scripts/traffic_merge.py r04h gpurun_out/final0/names.log | grep -E "^cfg"
FINAL_SKIP_TRAFFIC=1 bash scripts/gpu_r4_final.sh
cp profiles/traffic_latest.json profiles/r04h_pmc_hbm_traffic.txt gpurun_out/final/ 2>/dev/null
cp gpurun_out/final0/*.log gpurun_out/final/ 2>/dev/null
