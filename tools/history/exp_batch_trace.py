"""How do B windows optimised side by side (sdvgn_ef_optimize_batch, own stream per handle) share the device?

    python tools/exp_batch_trace.py run B            the workload: B windows, warm-up, 3 timed batch calls (prints ms per call)
    python tools/exp_batch_trace.py trace B [ENV=V]  the same under rocprofv3 --kernel-trace, then a timeline analysis of the last call:
                                                    wall, sum of kernel durations, time with >= 1 / >= 2 kernels running, per-queue
                                                    listing of the first kernels (start / end relative to the call's first kernel)
"""
import glob
import os
import sqlite3
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(B):
    import importlib
    import torch
    syn = importlib.import_module("sdv-loam_amd.synthetic")
    backend_api = importlib.import_module("sdv-loam_amd.backend_api")
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)
    stream = backend_api.EnergyFunctional.STREAM_OWN if os.environ.get("EXP_SHARED_STREAM", "0") != "1" else None
    hs = [backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP, device=0, stream=stream).load(W) for _ in range(B)]
    backend_api.optimize_batch(hs, 6, fixed_its=True)
    for rep in range(3):
        for h in hs:
            h.load(W)
        torch.cuda.synchronize()
        time.sleep(0.002)
        t0 = time.perf_counter()
        backend_api.optimize_batch(hs, 6, fixed_its=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("B=%d call %d: %.3f ms, %.0f bodies/s aggregate" % (B, rep, 1e3 * dt, 6 * B / dt), flush=True)


def analyse(db, B):
    con = sqlite3.connect(db)
    names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view') and name like '%kernel%'")]
    print("kernel tables/views:", names)
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    print("columns of kernels:", cols)
    qcol = "queue_id" if "queue_id" in cols else ("queue" if "queue" in cols else None)
    scol = "stream_id" if "stream_id" in cols else ("stream" if "stream" in cols else None)
    sel = "select name, start, end%s%s from kernels order by start" % ((", " + qcol) if qcol else "", (", " + scol) if scol else "")
    rows = con.execute(sel).fetchall()
    print("kernel records:", len(rows))
    # the last batch call = the kernels after the last gap > 1.5 ms
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 1.5e6:
            cut = i
    call = rows[cut:]
    # drop the load() kernels/copies that precede the call: start at the first k_ef_linearize after the cut
    first = next((i for i, r in enumerate(call) if "k_ef_linearize" in r[0]), 0)
    call = call[first:]
    t0 = call[0][1]
    t1 = max(r[2] for r in call)
    ev = []
    for r in call:
        ev.append((r[1], 1))
        ev.append((r[2], -1))
    ev.sort()
    busy1 = busy2 = busy3 = 0
    depth = 0
    last = ev[0][0]
    for t, d in ev:
        if depth >= 1:
            busy1 += t - last
        if depth >= 2:
            busy2 += t - last
        if depth >= 3:
            busy3 += t - last
        depth += d
        last = t
    tot = sum(r[2] - r[1] for r in call)
    print("last call: %d kernels, wall %.1f us, sum of durations %.1f us, >=1 running %.1f us, >=2 running %.1f us, >=3 running %.1f us"
          % (len(call), (t1 - t0) / 1e3, tot / 1e3, busy1 / 1e3, busy2 / 1e3, busy3 / 1e3))
    by = {}
    for r in call:
        k = r[0].split("(")[0][-40:]
        a = by.setdefault(k, [0, 0.0])
        a[0] += 1
        a[1] += r[2] - r[1]
    for k, a in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print("  %-42s %5d launches  avg %8.2f us" % (k, a[0], a[1] / a[0] / 1e3))
    print("timeline of the first 90 kernels of the call (us from its first kernel): start end dur queue stream name")
    for r in call[:90]:
        q = r[3] if len(r) > 3 else -1
        s = r[4] if len(r) > 4 else -1
        print("  %9.2f %9.2f %7.2f  q%-4s s%-4s %s" % ((r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3, q, s, r[0].split("(")[0][-44:]))


def main():
    mode = sys.argv[1]
    B = int(sys.argv[2])
    if mode == "run":
        run(B)
        return
    env = dict(os.environ)
    tag = "B%d" % B
    for kv in sys.argv[3:]:
        k, v = kv.split("=", 1)
        env[k] = v
        tag += "_" + kv.replace("=", "")
    env["TMPDIR"] = "/tmp"
    out = "/tmp/exp_batch_" + tag
    subprocess.run(["rm", "-rf", out])
    cmd = ["rocprofv3", "--kernel-trace", "-d", out, "-o", "t", "--", sys.executable, os.path.abspath(__file__), "run", str(B)]
    p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    print("\n".join(l for l in p.stdout.splitlines() if l.startswith("B=")))
    dbs = glob.glob(out + "/**/*.db", recursive=True)
    if not dbs:
        print(p.stdout[-3000:])
        raise SystemExit("no trace database")
    analyse(dbs[0], B)


if __name__ == "__main__":
    main()
