# round 4: builds of attn.hip's split mode that differ by -D switches, under three concurrent clip groups (determinism: every repetition must
# print the same checksum and an empty list), then per-step time of each:  bash scripts/gpu_r4_attn_bisect.sh name:-Dx=1 ...
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_base.so
names="base"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  touch said_amd/csrc/attn.hip said_amd/csrc/tgemm.hip
  SAID_EXTRA_DEFS="$defs" python -m said_amd.build > gpurun_out/r4/ab_build_$name.log 2>&1; echo "build $name ($defs) exit=$?"
  cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_$name.so
  names="$names $name"
done
for n in $names; do
  echo "== $n"; SAID_AB_LIB=said_amd/lib/ab_$n.so timeout 200 python scripts/attn_split_det.py 1 32 ${DET_GROUPS:-0} 2>&1 | grep attn_split | cut -c1-200
  if [ -z "$DET_NO_INDEP" ]; then SAID_AB_LIB=said_amd/lib/ab_$n.so timeout 300 python scripts/attn_split_indep.py 1 11 3 2>&1 | grep attn_split | cut -c1-200; fi
done | tee gpurun_out/r4/attn_bisect.txt
for rep in 1 2; do for n in $names; do
  echo -n "$n: "; timeout 200 python scripts/ab_libs.py said_amd/lib/ab_$n.so 32 50 fp32 2>/dev/null | tail -1 | sed 's/.*so B/B/'
  echo -n "$n: "; timeout 200 python scripts/ab_libs.py said_amd/lib/ab_$n.so 1 200 fp32 2>/dev/null | tail -1 | sed 's/.*so B/B/'
done; done | tee gpurun_out/r4/attn_bisect_time.txt
