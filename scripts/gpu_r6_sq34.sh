# round 6: SQ / LDS / L2 counters of the step kernels of the headline and configs[2] on the final sources (separate --pmc passes, kernel-trace only:
# MI355X_MICROARCH.md "rocprofv3 PMC slots").  Pass A: wave-cycle breakdown + MFMA busy; pass B: LDS bank conflicts, VALU / MFMA instruction counts; pass C: L2 hit / miss.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/sq6
rocprofv3 -L 2>/dev/null | grep -o -E "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+_sum|TCP_[A-Z0-9_]+_sum|GRBM_GUI_ACTIVE)\b" | sort -u > gpurun_out/sq6/available.txt
wc -l gpurun_out/sq6/available.txt
pick() { for c in "$@"; do grep -qx "$c" gpurun_out/sq6/available.txt && printf "%s " "$c"; done; }
A=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE)
B=$(pick SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS)
C=$(pick TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum)
echo "A: $A"; echo "B: $B"; echo "C: $C"
run() {  # name, bench flags
  name=$1; shift
  for p in A B C; do
    eval "set_=\$$p"
    [ -z "$set_" ] && continue
    timeout 500 rocprofv3 --pmc $set_ --kernel-trace -d gpurun_out/sq6 -o ${name}_$p -- python bench.py --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary "$@" > gpurun_out/sq6/run_${name}_$p.log 2>&1
    echo "$name $p exit=$?"
    db=$(find gpurun_out/sq6 -name "${name}_${p}_results.db" | head -1)
    python scripts/pmc_generic_summary.py $db said:: > gpurun_out/sq6/${name}_$p.txt 2>&1
  done
}
run cfg3_f32 --batch 32 --num_steps 10
run cfg4_edit --seconds 30 --num_steps 20 --edit
find gpurun_out/sq6 -name "*.db" -delete
head -12 gpurun_out/sq6/cfg3_f32_A.txt | cut -c1-260
