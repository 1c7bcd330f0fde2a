"""Profiling experiment (not product): k_track cycle breakdown (SDVGN_PROFILE=1) and call times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SDVGN_PROFILE"] = "1"
import numpy as np
import oracle
from sdv_loam_amd import api, synthetic as syn
P = syn.make_tracker_problem(1241, 376, 4, 2000, seed=0, calib=syn.KITTI00, gt_xi=[0.1, -0.05, 0.2, 0.01, -0.02, 0.005], gt_aff=(0.05, 3.0))
rng = np.random.default_rng(9)
for r in P.ref:
    r["color"] = (r["color"] + rng.normal(0, 1.0, r["color"].shape)).astype(np.float32)
G = api.CoarseTracker(P.w, P.h, P.levels, max_points=4096, max_batch=64)
G.makeK(**P.calib)
for l in range(P.levels):
    G.set_ref(l, **P.ref[l])
G.set_ref_frame(1.0, 0.0, 0.0)
G.set_new_image(P.image, 1.0)
for B in (1, 31):
    starts = np.stack([oracle.se3_mul(oracle.se3_exp(syn.perturbation(i)), P.gt_pose) for i in range(B)])
    affs = np.tile([0.02, 2.0], (B, 1))
    G.trackBatch(starts, affs, 3)
    t0 = time.perf_counter()
    for _ in range(10):
        G.trackBatch(starts, affs, 3)
    print("B=%d  %.3f ms/call" % (B, 1e2 * (time.perf_counter() - t0)), flush=True)
