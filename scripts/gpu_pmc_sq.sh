# SQ counters of the denoise-step kernels (one pass, kernel-trace only), per MI355X_MICROARCH.md §rocprofv3 PMC slots
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmcsq; mkdir -p gpurun_out/pmcsq
rocprofv3 -L 2>/dev/null | grep -i -E "MFMA|SQ_WAVE_CYCLES|SQ_BUSY_CYCLES|SQ_WAIT_ANY|SQ_WAIT_INST_ANY|SQ_ACTIVE_INST_ANY|GRBM_GUI_ACTIVE" | cut -c1-160 | head -30 > gpurun_out/pmcsq/list.txt
head -30 gpurun_out/pmcsq/list.txt
timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmcsq -o sq -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline > gpurun_out/pmcsq/run.log 2>&1; echo "exit=$?" >> gpurun_out/pmcsq/run.log; tail -2 gpurun_out/pmcsq/run.log | cut -c1-200
ls -la gpurun_out/pmcsq
