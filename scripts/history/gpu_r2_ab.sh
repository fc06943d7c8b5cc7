# round 2: parity suite on the new build, default bench, A/B of the round's schedule changes, large-batch lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/t2.log 2>&1; echo exit=$? >> gpurun_out/t2.log; tail -4 gpurun_out/t2.log
grep -E "max abs err|max err|bf16|g9 |teacher|vae|FAILED|Error" gpurun_out/t2.log | head -60
B="python bench.py --steps 3 --warmup 1 --no_cpu_baseline"
timeout 300 $B > gpurun_out/ab_default.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|graph_nodes_per_step": [0-9]*\|"ms_per_clip": [0-9.]*' gpurun_out/ab_default.log | tr '\n' ' '; echo " <- default"
SAID_NO_FFPROJ_FOLD=1 timeout 300 $B --no_roofline > gpurun_out/ab_nofold.log 2>&1; grep -o '"value": [0-9.]*\|graph_nodes_per_step": [0-9]*' gpurun_out/ab_nofold.log | tr '\n' ' '; echo " <- no ffproj fold"
SAID_NO_CFG_SHARE=1 timeout 300 $B --no_roofline > gpurun_out/ab_noshare.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/ab_noshare.log | tr '\n' ' '; echo " <- no cfg share"
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L > gpurun_out/b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*\|"mfma_frac": [0-9.]*' gpurun_out/b32_f32.log | tr '\n' ' '; echo " <- B=32 f32"
SAID_NO_CFG_SHARE=1 SAID_NO_FFPROJ_FOLD=1 timeout 300 $L --no_roofline > gpurun_out/b32_f32_old.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/b32_f32_old.log | tr '\n' ' '; echo " <- B=32 f32, round-1 schedule"
timeout 300 $L --dtype bf16 > gpurun_out/b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"mfma_frac": [0-9.]*' gpurun_out/b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16"
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary.txt 2>&1; head -30 gpurun_out/prof_summary.txt
