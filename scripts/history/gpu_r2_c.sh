# round 2, third batch: tuned chain kernel A/B, bf16 audio encoder parity + speed, two-workgroups-per-CU experiment
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "unet_forward_vs or loop_cfg_1s or g9 or bf16_audio or audio_encoder" > gpurun_out/t4a.log 2>&1; echo exit=$? >> gpurun_out/t4a.log; tail -3 gpurun_out/t4a.log | cut -c1-300
grep -E "bf16 audio|audio .*max abs" gpurun_out/t4a.log
B="python bench.py --steps 3 --warmup 1 --no_cpu_baseline"
timeout 300 $B > gpurun_out/c_default.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|graph_nodes_per_step": [0-9]*\|"xattn_kernel": {[^}]*}' gpurun_out/c_default.log | tr '\n' ' '; echo " <- default (xattn)"
SAID_NO_XATTN=1 timeout 300 $B --no_roofline > gpurun_out/c_off.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/c_off.log | tr '\n' ' '; echo " <- SAID_NO_XATTN=1"
timeout 300 $B --no_roofline --batch 2 --num_steps 100 > gpurun_out/c_b2.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/c_b2.log | tr '\n' ' '; echo " <- B=2 xattn"
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50"
timeout 300 $L --dtype bf16 > gpurun_out/c_b32_bf16.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*' gpurun_out/c_b32_bf16.log | tr '\n' ' '; echo " <- B=32 bf16 (bf16 audio)"
SAID_NO_AUDIO_BF16=1 timeout 300 $L --dtype bf16 --no_roofline > gpurun_out/c_b32_bf16_fa.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/c_b32_bf16_fa.log | tr '\n' ' '; echo " <- B=32 bf16, fp32 audio"
SAID_BIG_NB=1 timeout 300 $L --dtype bf16 > gpurun_out/c_b32_bf16_nb1.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/c_b32_bf16_nb1.log | tr '\n' ' '; echo " <- B=32 bf16 SAID_BIG_NB=1"
timeout 300 $L > gpurun_out/c_b32_f32.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/c_b32_f32.log | tr '\n' ' '; echo " <- B=32 f32"
SAID_BIG_NB=1 timeout 300 $L > gpurun_out/c_b32_f32_nb1.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*' gpurun_out/c_b32_f32_nb1.log | tr '\n' ' '; echo " <- B=32 f32 SAID_BIG_NB=1"
SAID_BIG_NB=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "batch32 or large_batch" > gpurun_out/t4b.log 2>&1; echo exit=$? >> gpurun_out/t4b.log; tail -3 gpurun_out/t4b.log | cut -c1-300
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 50 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary_b32_bf16.txt 2>&1; head -40 gpurun_out/prof_summary_b32_bf16.txt
