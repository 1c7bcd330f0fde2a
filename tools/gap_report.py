#!/usr/bin/env python3
"""Timeline of the main stream inside the optimize loops of the headline protocol: every kernel of the LAST traced optimize(6) call with its start relative to
the previous kernel's end (the gap the host / the dispatcher leaves), from a rocprofv3 --kernel-trace database.
    cd /tmp && rocprofv3 --kernel-trace -d /tmp/gp -o g -- python /root/repo/tools/bench_children.py trace ; python tools/gap_report.py /tmp/gp"""
import glob
import sqlite3
import sys

db = glob.glob(sys.argv[1] + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
want = [c for c in ("name", "start", "end", "stream_id", "queue_id") if c in cols]
rows = con.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
ix = {c: i for i, c in enumerate(want)}
# the main stream = the one k_ef_linearize runs on
names = [r[ix["name"]] for r in rows]
qkey = "stream_id" if "stream_id" in ix else ("queue_id" if "queue_id" in ix else None)
lin = [r for r in rows if "k_ef_linearize" in r[ix["name"]]]
mainq = lin[-1][ix[qkey]] if qkey else None
main = [r for r in rows if qkey is None or r[ix[qkey]] == mainq]
# the last call = the kernels after the last-but-one k_ef_stats_select pair ... simpler: the last 40 kernels of the main stream
tail = main[-(int(sys.argv[2]) if len(sys.argv) > 2 else 44):]
t0 = tail[0][ix["start"]]
prev_end = None
tot_gap = tot_run = 0.0
for r in tail:
    st, en = r[ix["start"]], r[ix["end"]]
    gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.2f us  gap %6.2f  run %6.2f  %s" % ((st - t0) / 1e3, gap, (en - st) / 1e3, r[ix["name"]].split("(")[0][-40:]))
    if prev_end is not None:
        tot_gap += max(gap, 0.0)
    tot_run += (en - st) / 1e3
    prev_end = max(en, prev_end or en)
print("span %.1f us; kernels %.1f us; gaps %.1f us" % ((prev_end - t0) / 1e3, tot_run, tot_gap))
side = [r for r in rows if qkey is not None and r[ix[qkey]] != mainq and r[ix["start"]] >= t0]
print("other streams in that span: %d kernels, e.g. %s" % (len(side), ", ".join(sorted({r[ix["name"]].split("(")[0][-24:] for r in side}))[:200]))
