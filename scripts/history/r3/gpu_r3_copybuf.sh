# where do the __amd_rocclr_copyBuffer dispatches of a headline run come from?  (kernel trace + memory-copy trace, no counters)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/cb; mkdir -p gpurun_out/cb
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace -d gpurun_out/cb -o t -- python bench.py --steps 1 --warmup 0 --num_steps 40 --no_cpu_baseline --no_roofline --no_secondary > gpurun_out/cb/run.log 2>&1
echo "exit=$?"
python - <<'PY'
import sqlite3, glob, collections, re
f = glob.glob('gpurun_out/cb/**/t_results.db', recursive=True)[0]
con = sqlite3.connect(f)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'copy' in t.lower() or 'kernel' in t.lower() or 'region' in t.lower()][:20])
rows = con.execute("select name, start, end, grid_x, workgroup_x, queue_id, stream_id from kernels order by start").fetchall()
cb = [r for r in rows if 'copyBuffer' in r[0]]
print(len(rows), 'kernels', len(cb), 'copyBuffer; grids:', collections.Counter((r[3], r[4]) for r in cb).most_common(5), 'queues:', collections.Counter(r[5] for r in cb), collections.Counter(r[5] for r in rows if 'ugemm' in r[0]))
# position of copyBuffer dispatches relative to out_sched (end of step)
last = None; seq = []
for r in rows:
    n = re.sub(r"\(.*", "", r[0])
    tag = 'CB' if 'copyBuffer' in n else ('OS' if 'out_sched' in n else ('CI' if 'conv_in' in n else None))
    if tag: seq.append(tag)
s = ''.join({'CB': 'c', 'OS': 'O', 'CI': 'I'}[t] for t in seq)
print(s[:400])
try:
    mc = con.execute("select * from memory_copies limit 3").fetchall(); print('memory_copies sample', mc)
    print(con.execute("select count(*) from memory_copies").fetchall())
except Exception as e: print('no memory_copies', e)
try:
    hip = con.execute("select name, count(*) from regions group by name order by 2 desc limit 25").fetchall(); print(hip)
except Exception as e: print('no regions', e)
PY
find gpurun_out/cb -name "*.db" -delete
