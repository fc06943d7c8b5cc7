cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# bf16 audio encoder: positional convolution as 16 bf16 GEMMs vs the grouped fp32 kernel
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "bf16_audio" > gpurun_out/t20.log 2>&1; echo exit=$? >> gpurun_out/t20.log; grep -a "audio\|passed\|failed\|Error\|error" gpurun_out/t20.log | tail -8 | cut -c1-200
L="python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch 32 --num_steps 50 --dtype bf16"
timeout 300 $L > gpurun_out/ac1.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ac1.log | tr '\n' ' '; echo " <- posconv on tgemm"
SAID_NO_POSCONV_TGEMM=1 timeout 300 $L > gpurun_out/ac0.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|"ms_per_clip": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/ac0.log | tr '\n' ' '; echo " <- posconv on the grouped fp32 kernel"
