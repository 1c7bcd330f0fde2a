#!/usr/bin/env python3
"""Phase timeline of the device-side small solve (backend_solve.inc) from wall_clock64() stamps of lane 0 of the solve workgroup.

usage (GPU box):  SDVGN_DEBUG_FLAGS=64 python tools/exp_solve_stamps.py [nF] [pts_per_kf]
Stamps: 0 workgroup 0 of k_ef_stitch starts | 1 k_ef_tail_resub starts | 2 H, b assembled from the shares | 3 blocked LDL^T done | 4 back substitution +
null-space projection done | 5 resubstitute inputs, step and precalc table written.  wall_clock64() ticks are 10 ns (constant 100 MHz);
8 / 9 / 10: workgroup 0 after its sums / products / shares.
Stamp 0 comes from another workgroup (possibly another XCD: the clocks agree to well under a microsecond)."""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
os.environ.setdefault("SDVGN_DEBUG_FLAGS", "64")
import torch  # noqa: E402,F401
from sdv_loam_amd import backend_api, synthetic as syn  # noqa: E402

nF = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ppk = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
W = syn.make_window(w=1241, h=376, nF=nF, pts_per_kf=ppk, seed=0, calib=syn.KITTI00)
G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP, stream=(backend_api.EnergyFunctional.STREAM_OWN if os.environ.get("EXP_OWN_STREAM") else None)).load(W)   # EXP_OWN_STREAM=1: the factorisation workgroup in a launch of its own (no parked workgroups beside it)
names = ["k_ef_stitch start -> k_ef_tail_resub start", "H, b assembly from the shares", "blocked LDL^T",
         "back substitution + orthogonalize", "xAd, step, precalc table"]
rows, sub, blk, stp = [], [], [], []
for rep in range(12):
    G.load(W)
    G.optimize(6, fixed_its=True)
    buf = np.zeros(16, np.uint64)
    n = G.L.sdvgn_debug_solve_stamps(G.h_, buf.ctypes.data_as(C.c_void_p))
    assert n == 16, "diagnostics off (SDVGN_DEBUG_FLAGS bit6)"
    st = buf[:6].astype(np.int64)
    rows.append(np.diff(st) / 100.0)
    b = buf.astype(np.int64)
    sub.append((np.array([b[8], b[9], b[10]]) - st[0]) / 100.0)
    blk.append(np.array([b[14] - b[2], b[15] - b[14], b[7] - b[15]]) / 100.0)
    stp.append(np.array([b[11] - st[5], b[12] - b[11], b[13] - b[12], b[13] - st[1]]) / 100.0)
rows = np.array(rows[2:])
print("device-side solve, nF = %d, %d points per key-frame: phase durations of the last body of %d optimize(6) calls (us)" % (nF, ppk, len(rows)))
print("%-44s %8s %8s %8s" % ("phase", "median", "min", "max"))
for k, nm in enumerate(names):
    print("%-44s %8.2f %8.2f %8.2f" % (nm, np.median(rows[:, k]), rows[:, k].min(), rows[:, k].max()))
print("%-44s %8.2f" % ("sum", np.median(rows.sum(axis=1))))
sub = np.array(sub[2:])
print("k_ef_stitch, workgroup 0 (host frame 0), since its start: accumulators in LDS %.2f us | products %.2f us | shares written %.2f us" % tuple(np.median(sub, axis=0)))
blk = np.array(blk[2:])
print("first LDL^T block: scaling + first panel %.2f us | barrier %.2f us | update + barrier %.2f us" % tuple(np.median(blk, axis=0)))
stp = np.array(stp[2:])
print("step workgroup: has its tagged words of x %.2f us after workgroup 0's last stamp | frame states (exp, compose, inverse) %.2f us | precalc table %.2f us | its last store %.2f us after k_ef_tail_resub's workgroup 0 started" % tuple(np.median(stp, axis=0)))
