#!/usr/bin/env python3
"""Wall time of one sdvgn_ef_optimize call of nb bodies (fixed count) on a freshly loaded window of the headline shape, nb = 1..6: is the cost linear in nb?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sdv_loam_amd import backend_api as api, synthetic as syn

Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)
hs = [api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(4)]
for h in hs:
    h.optimize(6, fixed_its=True, want_trace=False)
for nb in (1, 2, 3, 4, 5, 6, 2, 6):
    ts = []
    for rep in range(5):
        for h in hs:
            h.load(Wh)
        torch.cuda.synchronize()
        for h in hs:
            t0 = time.perf_counter()
            tr = h.optimize(nb, fixed_its=True, want_trace=True)
            ts.append((time.perf_counter() - t0) * 1e6)
    acc = "".join("A" if r[2] else "R" for r in tr)
    print("nb %d: median %.1f us  min %.1f  (last trace %s)" % (nb, np.median(ts), np.min(ts), acc))
