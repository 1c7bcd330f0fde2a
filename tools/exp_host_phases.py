"""Host wall time per phase of the optimize loop (SDVGN_PROFILE=1) on the headline window, default loop vs flags bit4 (no speculative re-solve).
usage (GPU box): SDVGN_PROFILE=1 python tools/exp_host_phases.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdv_loam_amd import backend_api, synthetic as syn

Wh = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)
Es = [backend_api.EnergyFunctional(Wh.w, Wh.h, max_points=Wh.nP).load(Wh) for _ in range(6)]
Es[0].optimize(6, fixed_its=True, want_trace=False)
for nospec in (False, True):
    for E in Es:
        E.load(Wh)
    sys.stderr.write("==== no_spec_solve=%s\n" % nospec)
    t0 = time.perf_counter()
    acc = []
    for E in Es:
        tr = E.optimize(6, fixed_its=True, want_trace=True, no_spec_solve=nospec)
        acc.append(tr[:, 2].astype(int))
    dt = time.perf_counter() - t0
    its = np.concatenate([E.iteration_times_us() for E in Es]).reshape(len(Es), -1)
    sys.stderr.write("accept patterns %s\nbody times (us) per call:\n%s\n%.1f us per body incl. per-call cost\n" % (acc[0], np.round(its, 1), 1e6 * dt / (6 * len(Es))))
