# round 4: bit-stability soak of the split-fp16 kernels (attn_kernel PM == 2, fgemm_kernel SP): one group, three groups, three independent models,
# 4 repetitions per process, several processes
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
for i in 1 2 3; do
  for g in 1 0 2; do timeout 200 python scripts/attn_split_det.py ${SOAK_ATTN:-1} 32 $g 2>&1 | grep attn_split | cut -c1-150; done
  timeout 300 python scripts/attn_split_indep.py ${SOAK_ATTN:-1} 20 3 2>&1 | grep attn_split | cut -c1-200
  timeout 300 python scripts/attn_split_indep.py ${SOAK_ATTN:-1} 11 3 2>&1 | grep attn_split | cut -c1-200
done | tee gpurun_out/r4/soak.txt
echo "distinct 32-clip checksums by group count:"; grep "B=32" gpurun_out/r4/soak.txt | sed 's/.*groups=\([0-9]\).*: \([0-9.]*\) nan.*/\1 \2/' | sort | uniq -c
echo "non-zero diffs among independent models:"; grep independent gpurun_out/r4/soak.txt | grep -v "\[0.0, 0.0, 0.0\]" | wc -l
