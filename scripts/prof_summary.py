"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel totals and one denoise step's timeline."""
import collections
import re
import sqlite3
import sys

db = sys.argv[1]
con = sqlite3.connect(db)
cur = con.cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, scratch_size "
                   "from kernels order by start").fetchall()
agg = collections.defaultdict(lambda: [0, 0.0])


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:90]


for r in rows:
    a = agg[short(r[0])]
    a[0] += 1
    a[1] += (r[2] - r[1]) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"{len(rows)} dispatches, {tot/1e3:.2f} ms of kernel time")
print(f"{'total ms':>10} {'calls':>7} {'avg us':>9} {'%':>6}  kernel")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"{v[1]/1e3:10.3f} {v[0]:7d} {v[1]/v[0]:9.2f} {100*v[1]/tot:6.1f}  {k}")
idx = [i for i, r in enumerate(rows) if "sched_step" in r[0] or "out_sched" in r[0]]   # last kernel of a denoise step
if len(idx) > 3:
    gaps = collections.Counter(b - a for a, b in zip(idx, idx[1:]))
    step_len = gaps.most_common(1)[0][0]
    k = next(k for k in range(len(idx) // 2, len(idx) - 1) if idx[k + 1] - idx[k] == step_len)
    i0, i1 = idx[k] + 1, idx[k + 1] + 1
    print(f"\none denoise step: {(rows[i1][1]-rows[i0][1])/1e3:.1f} us wall, {i1-i0} kernels, "
          f"{sum((r[2]-r[1]) for r in rows[i0:i1])/1e3:.1f} us in kernels")
    prev = None
    for r in rows[i0:i1]:
        gap = (r[1] - prev) / 1e3 if prev else 0.0
        print(f"  gap {gap:6.2f}  dur {(r[2]-r[1])/1e3:7.2f} us  grid {r[3]//r[6]}x{r[4]}x{r[5]} wg {r[6]} lds {r[7]} vgpr {r[8]}+{r[9]} sgpr {r[10]} scratch {r[11]}  {short(r[0])[6:70]}")
        prev = r[2]
