# scripts/ubench/bgemm.hip: bring-up of a direct-to-LDS bf16 GEMM on the audio encoder's projection shapes (+ knock-outs)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r6t28
for ko in 0 1 2 3 4 7; do
  hipcc --offload-arch=gfx950 -O3 -DVARIANT=0 -DKO=$ko -o /tmp/bgemm_$ko scripts/ubench/bgemm.hip 2>/dev/null && echo "== variant 0 knock-out $ko" | tee -a gpurun_out/r6t28/bgemm.txt && timeout 120 /tmp/bgemm_$ko | cut -c1-75 | tee -a gpurun_out/r6t28/bgemm.txt
done
