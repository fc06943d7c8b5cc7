# Round 4: SQ counters of the bf16 large-batch step's kernels (BASELINE configs[2]), one rocprofv3 --pmc pass (kernel trace only) + a plain
# kernel-trace pass of the same command for the un-instrumented durations.  bash scripts/gpu_r4_pmc_cfg2.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pmc4; mkdir -p gpurun_out/pmc4
CMD="python bench.py --batch 32 --num_steps 20 --dtype bf16 --steps 1 --warmup 0 --no_cpu_baseline --no_roofline --no_secondary"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace -d gpurun_out/pmc4 -o sq -- $CMD > gpurun_out/pmc4/run_sq.log 2>&1; echo "exit=$?" >> gpurun_out/pmc4/run_sq.log
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/pmc4 -o kt -- $CMD > gpurun_out/pmc4/run_kt.log 2>&1; echo "exit=$?" >> gpurun_out/pmc4/run_kt.log
ls gpurun_out/pmc4 | head -20
DB=$(ls gpurun_out/pmc4/*sq*.db 2>/dev/null | head -1)
python scripts/pmc_generic_summary.py $DB > gpurun_out/pmc4/sq_summary.txt 2>&1
KT=$(ls gpurun_out/pmc4/*kt*.db 2>/dev/null | head -1)
python scripts/prof_summary.py $KT > gpurun_out/pmc4/kt_summary.txt 2>&1
# (the rocpd databases are tens of MB each: gpurun merges at most 64 MiB back — keep the summaries only)
find gpurun_out/pmc4 -name '*.db' -delete; find gpurun_out/pmc4 -type d -empty -delete
head -40 gpurun_out/pmc4/sq_summary.txt; head -40 gpurun_out/pmc4/kt_summary.txt
