"""Per-kernel averages of whatever counters one rocprofv3 --pmc pass collected (rocpd sqlite): value summed over the counter's
instances, divided by the number of dispatches of that kernel.  usage: pmc_generic_summary.py <db> [name-substring ...]"""
import collections
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
want = sys.argv[2:]
rows = con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for k, c, v, d in rows:
    k = re.sub(r"\(.*", "", re.sub(r"^void ", "", k))
    if want and not any(w in k for w in want):
        continue
    agg[k][c] += float(v)
    disp[k].add(d)
names = sorted({c for a in agg.values() for c in a})
print("dispatches  " + "  ".join(f"{n:>26}" for n in names) + "  kernel")
for k in sorted(agg, key=lambda k: -len(disp[k])):
    n = max(len(disp[k]), 1)
    print(f"{n:10d}  " + "  ".join(f"{agg[k][c] / n:26.1f}" for c in names) + f"  {k[:60]}")
