#!/bin/bash
# build_variant.sh NAME "DEFS" SRC...: builds said_amd/lib/ab_NAME.so with extra -D switches applied to the named sources (everything else as shipped),
# then restores the shipped library.  Variant libraries travel to the GPU box with the snapshot; scripts load them through SAID_AB_LIB.
set -e
cd "$(dirname "$0")/.."
name=$1; defs=$2; shift 2
for s in "$@"; do touch said_amd/csrc/$s; done
SAID_EXTRA_DEFS="$defs" python -m said_amd.build > /tmp/build_$name.log 2>&1 || { tail -30 /tmp/build_$name.log; exit 1; }
cp said_amd/lib/libsaid_hip.so said_amd/lib/ab_$name.so
for s in "$@"; do touch said_amd/csrc/$s; done
python -m said_amd.build > /tmp/build_restore.log 2>&1 || { tail -30 /tmp/build_restore.log; exit 1; }
echo "built said_amd/lib/ab_$name.so ($defs)"
