# round 4: split-fp16 attention — key split (4 waves x one query tile) vs four query tiles per workgroup sharing K / V through LDS, on a
# development build (SAID_DEV_KNOBS: SAID_ATTN_KS), configs[4] and the headline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
cp said_amd/lib/libsaid_hip.so /tmp/ship.so
touch said_amd/csrc/engine.cpp
SAID_EXTRA_DEFS="-DSAID_DEV_KNOBS" python -m said_amd.build > gpurun_out/r4/devknobs_build.log 2>&1; echo "dev build exit=$?"
B="--no_cpu_baseline --no_roofline --no_secondary"
for rep in 1 2; do for ks in 0 -4 1 8; do
  echo -n "cfg4 SAID_ATTN_KS=$ks: "; SAID_ATTN_KS=$ks timeout 300 python bench.py --edit --seconds 30 --num_steps 100 --steps 3 --warmup 1 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r4/attn_qw.txt
for ks in 0 -4; do
  echo -n "cfg1 SAID_ATTN_KS=$ks: "; SAID_ATTN_KS=$ks timeout 300 python bench.py --steps 2 --warmup 1 $B 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done | tee -a gpurun_out/r4/attn_qw.txt
cp /tmp/ship.so said_amd/lib/libsaid_hip.so
