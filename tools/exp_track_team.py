#!/usr/bin/env python3
"""trackNewestCoarse on the device (sdvgn_tracker_track_batch) for team sizes -1 (k_track), 1, 2, 4, 8, 16, 32: call time and the
kernel's own cycle breakdown (SDVGN_PROFILE=1 prints it to stderr).   usage (GPU box): SDVGN_PROFILE=1 python tools/exp_track_team.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402,F401
import oracle  # noqa: E402
import bench  # noqa: E402
from sdv_loam_amd import api, synthetic as syn  # noqa: E402

P = bench.tracker_problem()
G = bench.load_tracker(api, P, 0, 64)
start = oracle.se3_mul(oracle.se3_exp(syn.perturbation(0)), P.gt_pose)
for B in (1, 31):
    starts = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(i)), P.gt_pose) for i in range(B)])
    affs = np.tile([0.02, 2.0], (B, 1))
    for team in (-1, 1, 2, 4, 8, 16, 32):
        G.set_team(team)
        G.trackBatch(starts, affs, 3)
        t0 = time.perf_counter()
        for _ in range(20):
            r = G.trackBatch(starts, affs, 3)
        dt = (time.perf_counter() - t0) / 20
        sys.stderr.flush()
        print("B %2d  team request %3d -> %2d   %.3f ms per call   pose[0] %s" % (B, team, G.last_team(), 1e3 * dt, np.array2string(r[1][0][:3], precision=9)), flush=True)
