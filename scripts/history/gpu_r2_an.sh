cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# configs[4] (30 s editing, B=1: 3600 tokens per launch): token-major path forced on vs the default channel-major kernels
for v in default 0; do
if [ $v = default ]; then unset SAID_UNET_TGEMM_MIN; else export SAID_UNET_TGEMM_MIN=$v; fi
timeout 400 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_roofline --seconds 30 --num_steps 100 --edit > gpurun_out/an.log 2>&1; echo "SAID_UNET_TGEMM_MIN=$v $(grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' gpurun_out/an.log | tr '\n' ' ')"
done
