# round 4: cache policy of rgemm's result stores (plain / sc1 write-through / nt), same box, one process per run
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_aux0.so
for aux in 16 2 17; do
  touch said_amd/csrc/rgemm.hip
  SAID_EXTRA_DEFS=-DSAID_RG_ST_AUX=$aux python -m said_amd.build > gpurun_out/r4/aux_build_$aux.log 2>&1; echo "build aux=$aux exit=$?"
  cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_aux$aux.so
done
for rep in 1 2; do for aux in 0 16 2 17; do
  echo -n "aux=$aux: "; timeout 200 python scripts/ab_libs.py said_amd/lib/ab_aux$aux.so 32 50 bf16 2>/dev/null | tail -1
done; done | tee gpurun_out/r4/staux_ab.txt
