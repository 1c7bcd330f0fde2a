#!/usr/bin/env python3
"""Duration of k_ef_linearize depending on what ran before it (rocprofv3 --kernel-trace of a child process).
usage (GPU box): python tools/exp_linearize_context.py > profiles/rNN_linearize_context.txt"""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
PATTERNS = [(0, 0, "alone, back to back"), (1, 0, "behind accumulate + reduce"), (2, 0, "behind stitch + tail + resubstitute"),
            (3, 10, "behind a one-wave kernel waiting 10 us"), (3, 30, "behind a one-wave kernel waiting 30 us"),
            (3, 100, "behind a one-wave kernel waiting 100 us"), (4, 0, "behind accumulate + reduce + stitch + tail + resubstitute"),
            (0, 0, "alone again")]
REPS = 30


def child():
    import torch  # noqa: F401
    import bench
    from sdv_loam_amd import backend_api, synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **bench.HEAD_KW)
    G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    G.optimize(2, fixed_its=True, want_trace=False)
    for pat, spin, _ in PATTERNS:
        G._check(G.L.sdvgn_debug_launch_pattern(G.h_, pat, REPS, spin))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child()
        return
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    d = tempfile.mkdtemp(prefix="sdvgn_ctx_", dir="/tmp")
    subprocess.run([exe, "--kernel-trace", "-d", d, "-o", "tr", "--", sys.executable, os.path.abspath(__file__), "--child"],
                   cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
    dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
    con = sqlite3.connect(dbs[0])
    rows = [r[0] for r in con.execute("select duration from kernels where name like '%k_ef_linearize%' order by start")]
    rows = np.array(rows[-REPS * len(PATTERNS):], np.float64) / 1e3
    print("# k_ef_linearize (cfg3 window), %d launches per line, rocprofv3 kernel durations in us" % REPS)
    for i, (_, _, name) in enumerate(PATTERNS):
        x = rows[i * REPS:(i + 1) * REPS][3:]
        print("%-62s mean %6.2f  median %6.2f  min %6.2f  max %6.2f" % (name, x.mean(), np.median(x), x.min(), x.max()))
    shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
