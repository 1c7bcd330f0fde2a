"""Kernel timeline of ONE sdvgn_ef_optimize call on the headline window (shared stream, default loop): start / end of every kernel relative to the
call's first kernel, with the gaps between them.   python tools/exp_call_timeline.py trace [flags]   (runs itself under rocprofv3 --kernel-trace)"""
import glob
import os
import sqlite3
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(nospec):
    import torch
    from sdv_loam_amd import backend_api, synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)
    hs = [backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for _ in range(3)]
    hs[0].optimize(6, fixed_its=True, want_trace=False)
    hs[0].load(W)
    torch.cuda.synchronize()
    time.sleep(0.003)
    for h in hs:
        tr = h.optimize(6, fixed_its=True, want_trace=True, no_spec_solve=nospec)
    torch.cuda.synchronize()
    print("accept", tr[:, 2].astype(int), "body us", [round(x, 1) for x in hs[-1].iteration_times_us()])


def analyse(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end, queue_id from kernels order by start").fetchall()
    cut = 0
    for i in range(1, len(rows)):
        if rows[i][1] - rows[i - 1][2] > 1.5e6:
            cut = i
    call = rows[cut:]
    # the LAST optimize call: from the last initial k_ef_linearize that follows a gap
    starts = [i for i, r in enumerate(call) if "k_ef_linearize" in r[0]]
    # 7 linearise launches per call
    first = starts[-7]
    call = call[first:]
    t0 = call[0][1]
    prev_end = t0
    for name, s, e, q in call:
        short = name.split("(")[0].replace("void sdvgn::", "").replace("sdvgn::", "")[:34]
        print("%-34s q%-2d start %7.1f  dur %6.1f  gap-before %6.1f" % (short, q, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = max(prev_end, e)
    print("call wall %.1f us" % ((max(r[2] for r in call) - t0) / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(len(sys.argv) > 2 and sys.argv[2] == "nospec")
    else:
        out = "/tmp/prof_timeline"
        subprocess.run(["rm", "-rf", out])
        env = dict(os.environ)
        subprocess.run(["rocprofv3", "--kernel-trace", "-d", out, "-o", "tl", "--", sys.executable, os.path.abspath(__file__), "run"] + sys.argv[2:],
                       cwd="/tmp", env=dict(env, TMPDIR="/tmp"))
        analyse(glob.glob(out + "/**/*.db", recursive=True)[0])
