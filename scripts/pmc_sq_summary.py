"""Per-kernel SQ counters from one rocprofv3 --pmc pass (rocpd sqlite): MFMA busy share and wave-cycle breakdown.

MfmaUtil here = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel duration x 2.4 GHz), with the duration taken from
profiles/ kernel trace of the un-instrumented run (passed as name=us pairs) because dispatches are serialised and
stretched under counter collection.  wait / instwait / active are shares of SQ_WAVE_CYCLES
(MI355X_MICROARCH.md: WAIT_ANY = parked on s_waitcnt/barrier, WAIT_INST_ANY = issue stall, ACTIVE_INST_ANY = issuing).

usage: pmc_sq_summary.py <sq.db> [kernel-substring=avg_us ...]
"""
import collections
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
dur = dict(a.rsplit("=", 1) for a in sys.argv[2:])
rows = con.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for k, c, v in rows:
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*", "", k)
    agg[k][c] += float(v)
    if c == "SQ_WAVE_CYCLES":
        cnt[k] += 1
print(f"{'launches':>8} {'MFMA_BUSY':>10} {'avg us':>7} {'MfmaUtil%':>9} {'wait%':>6} {'instwait%':>9} {'active%':>8}  kernel")
for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
    if "said::" not in k:
        continue
    a = agg[k]
    n = max(cnt[k], 1)
    mf = a["SQ_VALU_MFMA_BUSY_CYCLES"] / n
    wc = max(a["SQ_WAVE_CYCLES"], 1)
    us = next((float(v) for s, v in dur.items() if s in k), None)
    util = f"{100 * mf / (1024 * us * 2400):9.1f}" if us else "        -"
    print(f"{n:8d} {mf:10.0f} {us if us else 0:7.2f} {util} {100*a['SQ_WAIT_ANY']/wc:6.1f} {100*a['SQ_WAIT_INST_ANY']/wc:9.1f} "
          f"{100*a['SQ_ACTIVE_INST_ANY']/wc:8.1f}  {k[:64]}")
