# round 4: same-box A/B of builds of rgemm.hip that differ by -D switches (one process per run, alternating): bash scripts/gpu_r4_ab_defs.sh name:-Dx=1 ...
# ("base" = the shipped library); VARIANT_SRC (default rgemm.hip) is the source touched between builds
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
SRC=${VARIANT_SRC:-rgemm.hip}
cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_base.so
names="base"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  touch said_amd/csrc/$SRC
  SAID_EXTRA_DEFS="$defs" python -m said_amd.build > gpurun_out/r4/ab_build_$name.log 2>&1; echo "build $name ($defs) exit=$?"
  cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_$name.so
  names="$names $name"
done
for rep in 1 2 3; do for n in $names; do
  echo -n "$n: "; timeout 200 python scripts/ab_libs.py said_amd/lib/ab_$n.so ${AB_B:-32} ${AB_N:-50} ${AB_DT:-bf16} 2>/dev/null | tail -1 | sed 's/.*so B/B/'
done; done | tee gpurun_out/r4/ab_defs.txt
