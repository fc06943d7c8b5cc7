# fused chain kernel: parity subset first, then the suite, then A/B benches and a kernel trace
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -x -k "unet_forward or loop_cfg_1s or g9 or loop_editing_mask or no_guidance" > gpurun_out/t3a.log 2>&1; echo exit=$? >> gpurun_out/t3a.log; tail -4 gpurun_out/t3a.log | cut -c1-300
B="python bench.py --steps 3 --warmup 1 --no_cpu_baseline"
timeout 300 $B > gpurun_out/xa_default.log 2>&1; grep -o '"value": [0-9.]*\|ms_loop_per_step": [0-9.]*\|graph_nodes_per_step": [0-9]*\|"xattn_kernel": {[^}]*}' gpurun_out/xa_default.log | tr '\n' ' '; echo " <- default (xattn)"
SAID_NO_XATTN=1 timeout 300 $B --no_roofline > gpurun_out/xa_off.log 2>&1; grep -o '"value": [0-9.]*\|graph_nodes_per_step": [0-9]*' gpurun_out/xa_off.log | tr '\n' ' '; echo " <- SAID_NO_XATTN=1"
for bb in 2 4 8; do
timeout 300 $B --no_roofline --batch $bb --num_steps 100 > gpurun_out/xa_b$bb.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/xa_b$bb.log | tr '\n' ' '; echo " <- B=$bb xattn"
SAID_NO_XATTN=1 timeout 300 $B --no_roofline --batch $bb --num_steps 100 > gpurun_out/xa_b${bb}_off.log 2>&1; grep -o '"value": [0-9.]*' gpurun_out/xa_b${bb}_off.log | tr '\n' ' '; echo " <- B=$bb no xattn"
done
timeout 1500 python -m pytest tests -q -m gpu -s > gpurun_out/t3.log 2>&1; echo exit=$? >> gpurun_out/t3.log; tail -4 gpurun_out/t3.log | cut -c1-300
grep -E "FAILED|Error" gpurun_out/t3.log | head
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 1 --num_steps 200 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1); python scripts/prof_summary.py $f > gpurun_out/prof_summary.txt 2>&1; head -24 gpurun_out/prof_summary.txt
