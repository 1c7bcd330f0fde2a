#!/usr/bin/env python3
"""Phase stamps of k_ef_tail_resub's factorisation workgroup (backend_solve.inc, SOLVE_STAMP), for solves launched back to back (warm L2 / instruction
cache) and for the solves inside optimize() (the kernel's code last ran a whole body ago).   SDVGN_DEBUG_FLAGS=64 python tools/exp_tail_stamps.py"""
import os
import sys
import ctypes as C

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("SDVGN_DEBUG_FLAGS", "64")
import torch  # noqa: E402,F401
from sdv_loam_amd import backend_api, synthetic as syn  # noqa: E402

W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)


def stamps():
    buf = np.zeros(16, np.uint64)
    assert G.L.sdvgn_debug_solve_stamps(G.h_, buf.ctypes.data_as(C.c_void_p)) == 16
    return buf.astype(np.int64)


names = ["1->2 assembly", "2->14 scale, barrier, rows", "14->3 LDL^T", "3->15 column loads", "15->7 back substitution", "7->4 orthogonalize", "4->5 publish, statistics"]
order = [1, 2, 14, 3, 15, 7, 4, 5]


def row(b):
    return [(b[order[i + 1]] - b[order[i]]) / 100.0 for i in range(len(order) - 1)]


def step_row(b):
    return "frame-step workgroup relative to x published (stamp 4), us: x seen %.2f | states + SE(3) exp done %.2f | precalc table written %.2f" % (
        (b[11] - b[4]) / 100.0, (b[12] - b[4]) / 100.0, (b[13] - b[4]) / 100.0)


def stitch_row(b):
    return "k_ef_stitch, host 0 part 0 (us): 0->8 loads %.2f | 8->9 products (T1, TC, B = A_h D_h, (h,h) terms) %.2f | 9->10 shares %.2f" % (
        (b[8] - b[0]) / 100.0, (b[9] - b[8]) / 100.0, (b[10] - b[9]) / 100.0)


G.optimize(6, fixed_its=True)
rows = []
for rep in range(6):
    G.load(W)
    G.optimize(6, fixed_its=True)
    b_loop = stamps()
    rows.append(row(b_loop))
rows = np.array(rows[1:])
print("inside optimize(6), last body (us): " + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, np.median(rows, axis=0))) + " | sum %.2f" % np.median(rows.sum(axis=1)))
print("last body of that call: " + step_row(b_loop))
G.load(W)
G.linearize() if hasattr(G, "linearize") else None
for rep in range(5):
    G.solveSystemF(2, 0.1)
    if rep == 4:
        print(stitch_row(stamps()))
    r = row(stamps())
    print("solveSystemF #%d back to back (us): " % rep + " | ".join("%s %.2f" % (n, v) for n, v in zip(names, r)) + " | sum %.2f" % sum(r))
