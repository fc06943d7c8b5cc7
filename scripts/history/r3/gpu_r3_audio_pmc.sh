#!/bin/bash
# SQ counters of the bf16 audio encoder's GEMM kernels (one --pmc pass per counter group, kernel-trace only)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/apmc; mkdir -p gpurun_out/apmc
run() {  # tag, counters...
  tag=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace -d gpurun_out/apmc -o $tag -- python scripts/audio_trace.py 32 bf16 > gpurun_out/apmc/run_$tag.log 2>&1
  python scripts/pmc_generic_summary.py $(find gpurun_out/apmc -name "${tag}_results.db" | head -1) tgemm attn_kernel conv0 > gpurun_out/apmc_$tag.txt 2>&1
}
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run b SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16
run c SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_SALU SQ_INSTS_VALU
find gpurun_out/apmc -name "*.db" -delete
cat gpurun_out/apmc_a.txt gpurun_out/apmc_b.txt gpurun_out/apmc_c.txt
