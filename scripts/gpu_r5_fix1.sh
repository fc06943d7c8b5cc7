# round 5: after the packed-fp32 fix (NO_SLP build + split defaults ON): aggressor build beside the fixed victims, soak, suite, bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r5
echo "== localise, worst aggressor build (idle slots between the split attention's MFMAs), fixed victims" | tee gpurun_out/r5/fix1.txt
SAID_AB_LIB=said_amd/lib/ab_o2.so RACE_SPLIT=1 RACE_ATTN=1 RACE_FULL=40 timeout 900 python scripts/race_localise.py 3 22 600 4 9 14 2>&1 | grep -v "amdgpu.ids" | cut -c1-300 | tee -a gpurun_out/r5/fix1.txt
for lib in ab_o2 libsaid_hip; do for i in 1 2 3; do
  echo "== soak $lib #$i: three clip groups, split attention + split GEMMs" | tee -a gpurun_out/r5/fix1.txt
  SAID_AB_LIB=said_amd/lib/$lib.so DET_GEMM_SPLIT=1 timeout 300 python scripts/attn_split_det.py 1 32 3 2>&1 | grep attn_split | cut -c1-160 | tee -a gpurun_out/r5/fix1.txt
done; done
for i in 1 2; do SAID_AB_LIB=said_amd/lib/ab_o2.so DET_GEMM_SPLIT=1 timeout 300 python scripts/attn_split_indep.py 1 11 3 2>&1 | grep attn_split | cut -c1-200 | tee -a gpurun_out/r5/fix1.txt; done
echo "== suite" | tee -a gpurun_out/r5/fix1.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee -a gpurun_out/r5/fix1.txt
echo "== bench" | tee -a gpurun_out/r5/fix1.txt
timeout 900 python bench.py 2>&1 | tail -3 | tee gpurun_out/r5/bench_default.json | cut -c1-1500
