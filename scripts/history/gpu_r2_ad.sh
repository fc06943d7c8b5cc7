cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r02 -- python bench.py --steps 1 --warmup 0 --num_steps 2 --batch 32 --dtype bf16 --no_cpu_baseline --no_roofline > gpurun_out/prof/run.log 2>&1
f=$(find gpurun_out/prof -name "*_results.db" | head -1)
python - $f <<'PY'
import sqlite3, sys, re
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
# the audio encoder = everything before the first conv_in_kernel
out = []
for n, s, e, gx, wx in rows:
    n = re.sub(r"\(.*", "", re.sub(r"^void ", "", n))
    if "conv_in_kernel" in n: break
    out.append((n, (e - s) / 1e3, gx // wx))
tot = sum(t for _, t, _ in out)
print(f"audio encoder: {len(out)} launches, {tot/1e3:.2f} ms of kernel time")
import collections
agg = collections.OrderedDict()
for n, t, g in out:
    k = (n, g)
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += t
for (n, g), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{t/1e3:8.3f} ms {c:4d} x {t/c:9.1f} us  grid {g:6d}  {n[:70]}")
PY
find gpurun_out -name "*.db" -delete; rm -rf gpurun_out/prof
