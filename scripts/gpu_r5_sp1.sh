# round 5: split-fp16 products in the small-batch ugemm kernels (gemm_lds.hip SP): suite, then headline A/B (ugemm_split 1 / 0)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== suite" | tee gpurun_out/r5/sp1.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee -a gpurun_out/r5/sp1.txt
for v in 1 0 1 0; do
  echo "== bench ugemm_split=$v" | tee -a gpurun_out/r5/sp1.txt
  timeout 600 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_secondary --debug_option ugemm_split=$v 2>&1 | tail -1 | cut -c1-400 | tee -a gpurun_out/r5/sp1.txt
done
