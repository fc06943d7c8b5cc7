# SQ counters of the fp32 B=32 step: channel-major kernels (before) vs token-major fp32 GEMM (after); VERDICT r1 item 4
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for arm in after before; do
rm -rf gpurun_out/pmcsq; mkdir -p gpurun_out/pmcsq
if [ $arm = before ]; then export SAID_NO_UNET_FGEMM=1; else unset SAID_NO_UNET_FGEMM; fi
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcsq -o sq -- python bench.py --steps 1 --warmup 0 --num_steps 10 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/pmcsq/run.log 2>&1; echo "exit=$?" >> gpurun_out/pmcsq/run.log; tail -1 gpurun_out/pmcsq/run.log | cut -c1-100
f=$(find gpurun_out/pmcsq -name "*_results.db" | head -1)
python scripts/pmc_sq_summary.py $f > gpurun_out/pmc_sq_b32_f32_$arm.txt 2>&1
python - $f >> gpurun_out/pmc_sq_b32_f32_$arm.txt <<'PY'
import sqlite3, sys, collections, re
con = sqlite3.connect(sys.argv[1])
# per-dispatch: GRBM_GUI_ACTIVE / duration = effective clock; MFMA busy / (1024 x GUI_ACTIVE) = utilisation in CLOCKS (DVFS-free)
rows = con.execute("select kernel_name, counter_name, value, start, end from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for k, c, v, s, e in rows:
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", k)); agg[k][c] += float(v)
    if c == "GRBM_GUI_ACTIVE": agg[k]["ns"] += (e - s); n[k] += 1
print("\nlaunches   avg us(profiled)  eff GHz  MFMA busy / (1024 SIMD x GUI_ACTIVE clocks)  kernel")
for k in sorted(agg, key=lambda k: -agg[k]["ns"]):
    a = agg[k]
    if "said::" not in k or not a["GRBM_GUI_ACTIVE"]: continue
    print(f"{n[k]:8d} {a['ns']/n[k]/1e3:12.1f} {a['GRBM_GUI_ACTIVE']/a['ns']:12.2f} {100*a['SQ_VALU_MFMA_BUSY_CYCLES']/(1024*a['GRBM_GUI_ACTIVE']):10.1f} %   {k[:70]}")
PY
head -40 gpurun_out/pmc_sq_b32_f32_$arm.txt | cut -c1-170
done
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/pmcsq
