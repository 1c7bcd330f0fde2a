"""Shared helpers for the tracker tests: build the same problem on the GPU tracker and on the CPU oracle."""
import numpy as np


def small_problem(seed=0, n=400, w=256, h=192, levels=3, gt=True, noise=0.0, gt_aff=None, **kw):
    from sdv_loam_amd import synthetic as syn
    calib = dict(fx=220.0, fy=230.0, cx=127.3, cy=95.6)
    gt_xi = [0.04, -0.03, 0.06, 0.006, -0.004, 0.003] if gt else None
    P = syn.make_tracker_problem(w=w, h=h, levels=levels, n_points=n, seed=seed, calib=calib, gt_xi=gt_xi,
                                 gt_aff=gt_aff if gt_aff is not None else ((0.04, 2.5) if gt else (0.0, 0.0)), **kw)
    if noise > 0:
        rng = np.random.default_rng(seed + 77)
        for r in P.ref:
            r["color"] = (r["color"] + rng.normal(0, noise, r["color"].shape)).astype(np.float32)
    return P


def load_problem(T, P, ref_aff=(0.0, 0.0), exposures=(1.0, 1.0)):
    T.makeK(**P.calib)
    for l in range(P.levels):
        T.set_ref(l, **P.ref[l])
    T.set_ref_frame(exposures[0], ref_aff[0], ref_aff[1])
    T.set_new_image(P.image, exposures[1])
    return T


def start_pose(orc, P, seed=0, sigma_t=0.03, sigma_r=0.004):
    from sdv_loam_amd import synthetic as syn
    return orc.se3_mul(orc.se3_exp(syn.perturbation(seed, sigma_t, sigma_r)), P.gt_pose)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)
