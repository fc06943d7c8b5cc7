cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# the "nccl" (= RCCL) code path on a single-GPU box: one-rank process group, all-gather + barriers + max-reduce inside the sharded run
timeout 600 python bench.py --rccl_at_one --steps 2 --warmup 1 --num_steps 100 --no_cpu_baseline --no_roofline > gpurun_out/rccl_one.log 2>&1; echo exit=$? >> gpurun_out/rccl_one.log; tail -3 gpurun_out/rccl_one.log | cut -c1-1200
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --rccl_at_one --steps 2 --warmup 1 --num_steps 100 --batch 4 --no_cpu_baseline --no_roofline > gpurun_out/rccl_one_torchrun.log 2>&1; echo exit=$? >> gpurun_out/rccl_one_torchrun.log; tail -2 gpurun_out/rccl_one_torchrun.log | cut -c1-600
