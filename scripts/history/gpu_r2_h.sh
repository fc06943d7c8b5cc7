# attention with four query tiles per workgroup; counters for the large-batch bf16 kernels
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16 or batch32 or large_batch or audio_encoder" > gpurun_out/t9a.log 2>&1; echo exit=$? >> gpurun_out/t9a.log; tail -3 gpurun_out/t9a.log | cut -c1-300
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/h_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/h_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16 (QW attention)"
SAID_NO_ATTN_QW=1 timeout 300 $L --dtype bf16 --no_roofline > gpurun_out/h_b32_bf16_noqw.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/h_b32_bf16_noqw.log | tr '\n' ' '; echo " <- B=32 bf16 SAID_NO_ATTN_QW=1"
timeout 300 $L > gpurun_out/h_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/h_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32 (QW attention)"
rm -rf gpurun_out/pmcb; mkdir -p gpurun_out/pmcb
rocprofv3 -L 2>/dev/null | grep -i -E "^\s*(gpu-agent|.*Name).*(TCC_HIT|TCC_MISS|TCC_REQ|TCP_TCC_READ|LDS_BANK|SQ_INSTS_LDS|SQ_WAIT_INST_LDS|TCC_EA0_RDREQ|FETCH_SIZE|WRITE_SIZE|TCP_PENDING|TA_BUSY)" | cut -c1-200 | head -60 > gpurun_out/pmcb/list.txt
wc -l gpurun_out/pmcb/list.txt
P="python bench.py --steps 1 --warmup 0 --num_steps 10 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmcb -o sq -- $P > gpurun_out/pmcb/run_sq.log 2>&1; echo "sq exit=$?"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmcb -o fetch -- $P > gpurun_out/pmcb/run_f.log 2>&1; echo "fetch exit=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmcb -o write -- $P > gpurun_out/pmcb/run_w.log 2>&1; echo "write exit=$?"
timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d gpurun_out/pmcb -o tcc -- $P > gpurun_out/pmcb/run_t.log 2>&1; echo "tcc exit=$?"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmcb -o lds -- $P > gpurun_out/pmcb/run_l.log 2>&1; echo "lds exit=$?"
python scripts/pmc_sq_summary.py $(find gpurun_out/pmcb -name "sq*_results.db" | head -1) "tgemm_kernel<64>=41.5" "tgemm_kernel<128>=114" "attn_kernel<1, 1, true, 4>=100" "prep_kernel=17.7" > gpurun_out/pmcb/sq_summary.txt 2>&1; head -14 gpurun_out/pmcb/sq_summary.txt
python scripts/pmc_summary.py $(find gpurun_out/pmcb -name "fetch*_results.db" | head -1) $(find gpurun_out/pmcb -name "write*_results.db" | head -1) gpurun_out/pmcb/traffic_tgemm64.json "tgemm_kernel<64>" "tgemm_kernel<64>" > gpurun_out/pmcb/traffic_summary.txt 2>&1; head -14 gpurun_out/pmcb/traffic_summary.txt
python - <<'PY'
import sqlite3, glob, collections, re
for tag in ("tcc","lds"):
    fs = glob.glob(f"gpurun_out/pmcb/**/{tag}*_results.db", recursive=True)
    if not fs: print(tag, "no db"); continue
    con = sqlite3.connect(fs[0])
    try: rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    except Exception as e: print(tag, e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for k,c,v in rows:
        k = re.sub(r"\(.*","",re.sub(r"^void ","",k)); agg[k][c]+=float(v); cnt[(k,c)]+=1
    for k in sorted(agg, key=lambda k:-sum(agg[k].values()))[:8]:
        print(tag, k[:50], {c: round(v/max(cnt[(k,c)],1)) for c,v in agg[k].items()})
PY
