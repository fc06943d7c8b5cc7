cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "unet_forward_vs or loop_cfg_1s or g9" > gpurun_out/t5a.log 2>&1; echo exit=$? >> gpurun_out/t5a.log; tail -3 gpurun_out/t5a.log | cut -c1-300
B="python bench.py --steps 3 --warmup 1 --no_cpu_baseline"
timeout 300 $B > gpurun_out/d_default.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|graph_nodes_per_step": [0-9]*\|"xattn_kernel": {[^}]*}' gpurun_out/d_default.log | tr '\n' ' '; echo " <- default (xattn v3)"
SAID_NO_XATTN=1 timeout 300 $B --no_roofline > gpurun_out/d_off.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/d_off.log | tr '\n' ' '; echo " <- SAID_NO_XATTN=1"
