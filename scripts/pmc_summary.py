"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (rocpd sqlite) per kernel and write the
dominant kernel's HBM traffic per launch for bench.py (profiles/traffic_latest.json).

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section):
FETCH_SIZE and WRITE_SIZE count kilobytes; on gfx950 FETCH_SIZE reports half of the bytes of wide
(16 B per lane) coalesced reads, which is what these kernels issue, so reads are doubled.

usage: pmc_summary.py <fetch.db> <write.db> <out.json> [rocprof-kernel-substring] [bench-kernel-name]
"""
import collections
import json
import re
import sqlite3
import sys


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)).fetchall()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, v in rows:
        n = re.sub(r"^void ", "", name)
        n = re.sub(r"\(.*", "", n)
        a = agg[n]
        a[0] += 1
        a[1] += float(v)
    return agg


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
pick = sys.argv[4] if len(sys.argv) > 4 else "ugemm_kernel<1, 8, 0"
bench_name = sys.argv[5] if len(sys.argv) > 5 else "ugemm_kernel<NB1,KS8,store>"   # bench.py's name for the same kernel family
print(f"{'launches':>9} {'FETCH KB':>10} {'WRITE KB':>10} {'HBM MB (2*F+W)':>15}  kernel")
tot_n, tot_b = 0, 0.0
for k in sorted(fetch, key=lambda k: -fetch[k][1]):
    n, f = fetch[k]
    w = write.get(k, [n, 0.0])[1]
    fa, wa = f / n, w / max(write.get(k, [n])[0], 1)
    mb = (2 * fa + wa) * 1024 / 1e6
    print(f"{n:9d} {fa:10.1f} {wa:10.1f} {mb:15.2f}  {k[:80]}")
    if pick in k:
        tot_n += n
        tot_b += (2 * fa + wa) * 1024 * n
if tot_n:
    out = {"kernel": bench_name, "rocprof_match": pick, "launches": tot_n, "hbm_bytes_per_launch": round(tot_b / tot_n),
           "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024, launch-weighted over the kernel's variants; gfx950 half-count correction applied to reads"}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(out))
