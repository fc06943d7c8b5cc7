cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
# where does the token-major path start to pay?  step time by batch with the path forced on / off, both precisions
for dt in ${DTS:-f32 bf16}; do
for B in ${BS:-4 6 8 12}; do
for min in 0 1000000000; do
SAID_UNET_TGEMM_MIN=$min timeout 300 python bench.py --steps 2 --warmup 1 --no_cpu_baseline --batch $B --num_steps 50 --dtype $dt > gpurun_out/w.log 2>&1
python - <<PY
import json
s=open('gpurun_out/w.log').read()
d=json.loads(s[s.index('{"metric'):].splitlines()[0])
print('$dt B=$B token-major', 'on ' if $min == 0 else 'off', 'step ms', d['roofline']['unet_step']['ms_loop_per_step'], 'frames/s', d['value'])
PY
done; done; done
