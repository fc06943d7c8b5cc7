# HBM traffic (PMC FETCH_SIZE / WRITE_SIZE, separate passes, kernel-trace only) of the dominant UNet kernel of the headline and
# of every secondary bench configuration -> gpurun_out/traffic/<cfg>_{FETCH,WRITE}_SIZE_results.db
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/traffic; mkdir -p gpurun_out/traffic
python -c "import bench; print(bench.source_hash())" > gpurun_out/traffic/source_hash.txt   # of the sources MEASURED (this snapshot)
run() {  # name, bench flags
  name=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/traffic -o ${name}_$c -- python bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/traffic/run_${name}_$c.log 2>&1
    echo "$name $c exit=$?"
  done
}
run cfg1 --num_steps 40
run cfg2_bf16 --batch 32 --num_steps 10 --dtype bf16
run cfg3_per_gpu_f32 --batch 32 --num_steps 10
run cfg4_edit --seconds 30 --num_steps 20 --edit
find gpurun_out/traffic -name "*.db" | xargs ls -la | awk '{print $5, $9}'
# keep the merged-back payload small: only the counters table matters
for f in $(find gpurun_out/traffic -name "*_results.db"); do
  python - "$f" <<'PY'
import sqlite3, sys, collections, json, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])
for k, c, v in rows:
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", k))
    a = agg[(k, c)]; a[0] += 1; a[1] += float(v)
json.dump([{"kernel": k, "counter": c, "launches": n, "sum": s} for (k, c), (n, s) in agg.items()], open(sys.argv[1].replace("_results.db", "_summary.json"), "w"))
PY
done
find gpurun_out/traffic -name "*.db" -delete
find gpurun_out/traffic -type f | head -30
