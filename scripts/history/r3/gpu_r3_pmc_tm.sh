# SQ / LDS counters of the token-major activation path's kernels (B=32, bf16), separate --pmc passes with kernel-trace only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmctm; mkdir -p gpurun_out/pmctm
P="python bench.py --steps 1 --warmup 0 --num_steps 10 --batch 32 --dtype ${DT:-bf16} --no_cpu_baseline --no_roofline --no_secondary"
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d gpurun_out/pmctm -o sq -- $P > gpurun_out/pmctm/run_sq.log 2>&1; echo "sq exit=$?"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE --kernel-trace -d gpurun_out/pmctm -o lds -- $P > gpurun_out/pmctm/run_l.log 2>&1; echo "lds exit=$?"
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace -d gpurun_out/pmctm -o inst -- $P > gpurun_out/pmctm/run_i.log 2>&1; echo "inst exit=$?"
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/pmctm -o trace -- $P > gpurun_out/pmctm/run_t.log 2>&1; echo "trace exit=$?"
for t in sq lds inst; do
  f=$(find gpurun_out/pmctm -name "${t}*_results.db" | head -1)
  [ -n "$f" ] && python scripts/pmc_generic_summary.py $f xgemm attn_kernel prep tgemm fgemm > gpurun_out/pmctm/${t}_summary.txt 2>&1
done
python scripts/prof_summary.py $(find gpurun_out/pmctm -name "trace*_results.db" | head -1) > gpurun_out/pmctm/trace_summary.txt 2>&1
head -30 gpurun_out/pmctm/trace_summary.txt | cut -c1-180
cat gpurun_out/pmctm/sq_summary.txt gpurun_out/pmctm/lds_summary.txt gpurun_out/pmctm/inst_summary.txt | cut -c1-330
