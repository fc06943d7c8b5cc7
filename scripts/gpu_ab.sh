#!/bin/bash
# Same-box A/B of one engine change in ONE gpurun call (round 6's one driver; the 24 one-off gpu_r6_tNN.sh files it replaces are in `git log -- scripts/`):
#   bash scripts/gpurun_retry.sh 1800 "scripts/gpu_ab.sh CONFIG VARIANT [VARIANT ...]"
# CONFIG: headline | cfg2 | cfg3 | cfg4 | clipsN (N clips x 100 steps, fp32)
# VARIANT: NAME=VALUE (said_debug_option on the shipped library), lib:PATH (bench.py --ab_lib: a build_variant.sh library) or "base"; every variant runs twice, interleaved.
# With EQ=1 in front of CONFIG the first lib: variant is also checked for bit-identity against the shipped library (scripts/ab_equal.py).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
eq=0; if [ "$1" = "EQ=1" ]; then eq=1; shift; fi
cfg=$1; shift
case $cfg in
  headline) flags="--steps 5 --warmup 2";;
  cfg2) flags="--batch 32 --num_steps 50 --dtype bf16 --steps 3 --warmup 1";;
  cfg3) flags="--batch 32 --num_steps 100 --steps 2 --warmup 1";;
  cfg4) flags="--seconds 30 --num_steps 100 --edit --steps 3 --warmup 1";;
  clips*) flags="--batch ${cfg#clips} --num_steps 100 --steps 3 --warmup 1";;
  *) echo "unknown config $cfg"; exit 1;;
esac
out=gpurun_out/ab_$(date +%H%M%S); mkdir -p $out
if [ $eq = 1 ]; then
  for v in "$@"; do case $v in lib:*) timeout 600 python scripts/ab_equal.py said_amd/lib/libsaid_hip.so save /tmp/ref.pt | tail -1; timeout 600 python scripts/ab_equal.py ${v#lib:} cmp /tmp/ref.pt | tail -13 | tee $out/equal.txt; break;; esac; done
fi
for rep in 1 2; do for v in "$@"; do
  case $v in base) opt="";; lib:*) opt="--ab_lib ${v#lib:}";; *) opt="--debug_option $v";; esac
  echo "== $cfg $v" | tee -a $out/ab.txt
  timeout 900 python bench.py $flags --no_cpu_baseline --no_secondary --no_roofline $opt 2>&1 | tail -1 | cut -c1-160 | tee -a $out/ab.txt
done; done
