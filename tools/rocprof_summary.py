#!/usr/bin/env python3
"""Summarise a rocprofv3 sqlite result (`rocprofv3 --kernel-trace --stats -d DIR -- cmd`) as the text table kept under profiles/.

usage: python tools/rocprof_summary.py DIR_OR_DB "<command line that was profiled>" > profiles/rNN_rocprof_summary.txt
"""
import glob
import os
import sqlite3
import sys


def find_db(path):
    if os.path.isfile(path):
        return path
    dbs = sorted(glob.glob(os.path.join(path, "**", "*.db"), recursive=True), key=os.path.getmtime)
    if not dbs:
        raise SystemExit("no .db under %s" % path)
    return dbs[-1]


def main():
    db = find_db(sys.argv[1])
    cmd = sys.argv[2] if len(sys.argv) > 2 else "?"
    con = sqlite3.connect(db)
    cur = con.cursor()
    print("# rocprofv3 --kernel-trace --stats -- %s   (MI355X)" % cmd)
    print("# source: %s (view top_kernels); durations in microseconds" % os.path.basename(db))
    print("%-112s %7s %14s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    cols = [r[1] for r in cur.execute("pragma table_info(top_kernels)")]
    rows = cur.execute("select * from top_kernels").fetchall()
    ix = {c: i for i, c in enumerate(cols)}

    def col(r, *names):
        for n in names:
            if n in ix:
                return r[ix[n]]
        return None
    for r in rows:
        name = col(r, "name", "kernel_name")
        calls = col(r, "total_calls", "calls")
        tot = col(r, "total_duration (nsec)", "total_duration")
        avg = col(r, "average (nsec)", "average")
        pct = col(r, "percentage", "pct")
        print("%-112s %7d %14.1f %12.3f %8.2f" % (name[:112], calls, tot, avg, pct))
    print()
    print("# per-dispatch registers / LDS (kernels view)")
    try:
        kc = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        want = ["name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_x", "grid_x", "grid_y"]
        have = [w for w in want if w in kc]
        seen = set()
        for r in cur.execute("select %s from kernels" % ",".join(have)):
            d = dict(zip(have, r))
            if d["name"] in seen:
                continue
            seen.add(d["name"])
            print("%-90s vgpr=%4s agpr=%4s sgpr=%4s lds=%6s scratch=%5s wg=%5s grid=%s" % (
                d["name"][:90], d.get("vgpr_count"), d.get("accum_vgpr_count"), d.get("sgpr_count"), d.get("lds_size"),
                d.get("scratch_size"), d.get("workgroup_x"), "(%s,%s)" % (d.get("grid_x"), d.get("grid_y"))))
    except sqlite3.Error as e:
        print("# kernels view unavailable: %s" % e)
    # the roofline kernel by launch context: bench.py times it in back-to-back launches (the previous launch of the same kernel started
    # < 25 us earlier), the optimize loop launches it between other kernels
    try:
        rows = cur.execute("select start, duration from kernels where name like '%k_ef_linearize%' order by start").fetchall()
        if len(rows) > 2:
            st = [r[0] for r in rows]
            du = [r[1] / 1000.0 for r in rows]
            b2b = [du[i] for i in range(1, len(du)) if st[i] - st[i - 1] < 25000]
            loop = [du[i] for i in range(1, len(du)) if st[i] - st[i - 1] >= 25000] + du[:1]

            def stats(v):
                v = sorted(v)
                return "n=%d avg=%.2f median=%.2f p10=%.2f p90=%.2f us" % (len(v), sum(v) / len(v), v[len(v) // 2], v[len(v) // 10], v[(9 * len(v)) // 10]) if v else "n=0"
            print()
            print("# k_ef_linearize by launch context")
            print("#   back-to-back launches (what bench.py's roofline line times): %s" % stats(b2b))
            print("#   launches inside the optimize loop / API sequences:             %s" % stats(loop))
    except sqlite3.Error as e:
        print("# per-dispatch durations unavailable: %s" % e)


if __name__ == "__main__":
    main()
