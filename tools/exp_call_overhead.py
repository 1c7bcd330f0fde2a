#!/usr/bin/env python3
"""Fixed cost of one sdvgn_ef_optimize call on a fresh, HBM-resident window: optimize(nb) for nb = 0, 1, 2, 6 on windows that were loaded but
never optimised, and again on windows that were optimised once and reloaded.   usage (GPU box): python tools/exp_call_overhead.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402
import bench  # noqa: E402
from sdv_loam_amd import backend_api, synthetic as syn  # noqa: E402

Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, **bench.HEAD_KW)
R = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(8)]
R[0].optimize(6, fixed_its=True, want_trace=False)
R[0].load(Wh)
torch.cuda.synchronize()
for tag in ("never optimised", "optimised once + reloaded"):
    for nb in (0, 1, 2, 6):
        ts = []
        for r in R:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.optimize(nb, fixed_its=True, want_trace=False)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            ts.append((t1 - t0, t2 - t0))
            r.load(Wh)
        ts = np.array(ts) * 1e6
        print("%-26s optimize(%d): call returns after %6.1f us (median), stream drained after %6.1f us; first window %6.1f" % (
            tag, nb, np.median(ts[:, 0]), np.median(ts[:, 1]), ts[0, 1]))
# back to back, as the bench does: 8 calls of 6 bodies, one timed region
for r in R:
    r.load(Wh)
torch.cuda.synchronize()
t0 = time.perf_counter()
for r in R:
    r.optimize(6, fixed_its=True, want_trace=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
its = np.concatenate([r.iteration_times_us() for r in R])
print("8 calls x 6 bodies in one region: %.1f us per call, bodies median %.1f us -> fixed %.1f us per call" % (1e6 * dt / 8, np.median(its), 1e6 * dt / 8 - 6 * np.median(its)))
print("body times of the first call:", np.round(R[0].iteration_times_us(), 1))


def region(runners, bodies, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for r, nb in zip(runners, bodies):
        r.optimize(nb, fixed_its=True, want_trace=False)
        marks.append(time.perf_counter() - t0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = np.concatenate([r.iteration_times_us() for r in runners])
    print("%-44s total %7.1f us (calls returned at %s, drained +%.1f); bodies sum %.1f us" % (
        label, 1e6 * dt, np.round(1e6 * np.array(marks), 1), 1e6 * (dt - (t1 - t0)), its.sum()))


print("---- the bench's region: 4 calls [6, 6, 6, 2]")
fresh = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(4)]
warm = backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh)
warm.optimize(5, fixed_its=True, want_trace=False)
region(fresh, [6, 6, 6, 2], "4 handles that never ran optimize")
for r in fresh:
    r.load(Wh)
warm.optimize(5, fixed_its=True, want_trace=False)
region(fresh, [6, 6, 6, 2], "same handles, reloaded, warm-up before")
for r in fresh:
    r.load(Wh)
region(fresh, [6, 6, 6, 2], "same handles, reloaded, no warm-up")
for r in R[:4]:
    r.load(Wh)
warm.optimize(5, fixed_its=True, want_trace=False)
region(R[:4], [6, 6, 6, 2], "4 of the 8 older handles, warm-up before")
