# round 5: the floor of every launch of the headline step: all workgroups of every kernel but conv_in return at entry (ab_floor.so)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5; rm -rf gpurun_out/r5/tr_fl
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r5/tr_fl -o b1 -- python bench.py --steps 1 --warmup 1 --no_cpu_baseline --no_roofline --no_secondary --num_steps 200 --ab_lib said_amd/lib/ab_floor.so > gpurun_out/r5/run_fl.log 2>&1
python scripts/prof_summary.py $(find gpurun_out/r5/tr_fl -name "b1_results.db" | head -1) > gpurun_out/r5/floor_b1.txt 2>&1
rm -rf gpurun_out/r5/tr_fl
cut -c1-170 gpurun_out/r5/floor_b1.txt | sed -n 1,16p; grep -A30 "one denoise" gpurun_out/r5/floor_b1.txt | cut -c1-170
