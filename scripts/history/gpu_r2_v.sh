# memory-path counters of the fp32 token-major GEMMs (B=32 fp32 step): L2 read latency as seen by the CU's L1, L1 stalls, TLB, L2 hit rate
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_PENDING_STALL_CYCLES TCP_GATE_EN1" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCR_TCP_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES" "TCC_HIT TCC_MISS TCC_REQ TA_TA_BUSY"; do
i=$((i+1))
rm -rf gpurun_out/pmcm; mkdir -p gpurun_out/pmcm
timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcm -o m -- python bench.py --steps 1 --warmup 0 --num_steps 4 --batch 32 --no_cpu_baseline --no_roofline > gpurun_out/pmcm/run.log 2>&1; echo "exit=$?"
f=$(find gpurun_out/pmcm -name "*_results.db" | head -1)
python scripts/pmc_generic_summary.py $f fgemm attn_kernel prep_kernel > gpurun_out/pmc_mem_b32_f32_$i.txt 2>&1; cat gpurun_out/pmc_mem_b32_f32_$i.txt | cut -c1-200
done
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/pmcm
