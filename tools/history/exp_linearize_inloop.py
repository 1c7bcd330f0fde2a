#!/usr/bin/env python3
"""k_ef_linearize inside the optimize loops of the headline protocol, by position in the call (0 = the call's initial linearizeAll), for
8 windows with a handle each and for one handle reloaded 8 times.   usage (GPU box): python tools/exp_linearize_inloop.py"""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)


def child(mode):
    import torch  # noqa: F401
    import bench
    from sdv_loam_amd import backend_api, synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **bench.HEAD_KW)
    n = 8 if mode == "many" else 1
    R = [backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for _ in range(n)]
    R[0].optimize(6, fixed_its=True, want_trace=False)
    R[0].load(W)
    for i in range(8):
        r = R[i % n]
        r.optimize(6, fixed_its=True, want_trace=False)
        if mode != "many":
            r.load(W)
    torch.cuda.synchronize()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
        return
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    for mode in ("many", "one"):
        d = tempfile.mkdtemp(prefix="sdvgn_il_", dir="/tmp")
        subprocess.run([exe, "--kernel-trace", "-d", d, "-o", "tr", "--", sys.executable, os.path.abspath(__file__), "--child", mode],
                       cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
        con = sqlite3.connect(dbs[0])
        names = ["k_ef_linearize", "k_ef_acc_fused", "k_ef_acc_reduce", "k_ef_stitch", "k_ef_tail_resub", "k_ef_stats_select", "k_ef_stats_apply", "k_ef_apply", "k_ef_linearize"]
        print("# %s" % ("8 windows, a handle each" if mode == "many" else "one handle, reloaded before every call"))
        for nm in names:
            du = np.array([r[0] for r in con.execute("select duration from kernels where name like ? order by start", ("%" + nm + "%",))], np.float64) / 1e3
            if nm == "k_ef_linearize" and names.index(nm) == 0:
                du = du[-56:].reshape(8, 7)
                print("k_ef_linearize by position in the call (mean over 8 calls):", np.round(du.mean(axis=0), 2), " all: mean %.2f" % du.mean())
                print("   first call:", np.round(du[0], 2), " last call:", np.round(du[-1], 2))
            else:
                if not len(du):
                    continue
                k = len(du) * 8 // 9 if len(du) >= 9 else len(du)
                print("%-26s mean %6.2f  median %6.2f  (n %d)" % (nm, du[-k:].mean(), np.median(du[-k:]), k))
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
