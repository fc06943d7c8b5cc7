# round 5: reproduce round 4's frequent failure (attention split, issue order 2) and localise it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/race
for v in o2 o1; do
  echo "== $v: attn_split_det 3 groups"; SAID_AB_LIB=said_amd/lib/ab_$v.so timeout 300 python scripts/attn_split_det.py 1 32 3 2>&1 | grep attn_split | cut -c1-220
  echo "== $v: localise"; SAID_AB_LIB=said_amd/lib/ab_$v.so RACE_SPLIT=0 RACE_ATTN=1 RACE_FULL=20 timeout 900 python scripts/race_localise.py 3 22 600 6 2>&1 | grep -v "concurrent deviating  0 /" | cut -c1-400
  mv gpurun_out/race/events_split0_attn1.npz gpurun_out/race/events_$v.npz
done 2>&1 | tee gpurun_out/race/localise_attn.txt
