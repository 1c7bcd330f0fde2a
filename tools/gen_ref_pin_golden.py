#!/usr/bin/env python3
"""tests/golden/ref_pin*.npz = outputs of the REFERENCE's own code (oracle/_ref/libref.so: the reference's translation units compiled
unmodified, see oracle/Makefile and oracle/ref_glue*.cpp): ref_pin.npz for the input sets of oracle/refpin.py, ref_pin_backend.npz for one
synthetic window, ref_pin_tracker.npz for one synthetic tracking problem.  Run in the container that has /root/reference:
    make -C oracle ref && python tools/gen_ref_pin_golden.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import refpin  # noqa: E402



def backend_reference_outputs(cfg):
    """outputs of the REFERENCE's back end (RefEF: oracle/ref_glue_ef.cpp around the reference's own translation units) for one
    synthetic window -> dict of arrays (tests/golden/ref_pin_backend.npz; tests/test_ref_pin_backend.py)"""
    import sdv_loam_amd  # noqa: F401
    from oracle.backend import RefEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(**cfg)
    R = RefEF(W.w, W.h).load(W)
    R.compute_nullspaces()
    out = dict(lin_energy=np.float64(R.linearizeAll()), J_new=R.residual_J(0))
    st = R.residual_state()
    out.update(new_state=st["new_state"], new_energy=st["new_energy"])
    R.applyRes()
    R.solveSystemF(0, 0.1)
    out["top_acc"] = R.top_acc()
    for a, k in zip(R.sc_acc(), ("accE", "accEB", "accD", "Hcc", "bc")):
        out[k] = a
    out["points"] = R.points()
    out["x"] = R.system()["x"]
    R2 = RefEF(W.w, W.h).load(W)
    R2.compute_nullspaces()
    rmse, steps, removed, _ = R2.optimize_full(6)
    s = R2.state()
    out.update(opt_accept=np.array([a for a, _, _ in steps], np.uint8), opt_energy=np.array([e for _, _, e in steps]),
               opt_state=s[1], opt_idepth=s[2], opt_removed=removed, opt_rmse=np.float64(rmse))
    return out


def tracker_reference_outputs(cfg, start_seed=3):
    """outputs of the REFERENCE's coarse tracker (RefTracker: oracle/ref_glue_tracker.cpp) for one synthetic tracking problem"""
    import oracle
    import sdv_loam_amd  # noqa: F401
    from sdv_loam_amd import synthetic as syn
    P = syn.make_tracker_problem(**cfg)
    R = oracle.RefTracker(P.w, P.h, P.levels)
    R.makeK(**P.calib)
    for l in range(P.levels):
        R.set_ref(l, **P.ref[l])
    R.set_ref_frame(1.0, 0.01, 0.5)
    R.set_new_image(P.image, 1.0)
    start = oracle.ref_se3("mul", oracle.ref_se3("exp", syn.perturbation(start_seed, 0.02, 0.003)), P.gt_pose)
    out = dict(start=start)
    for l in range(P.levels):
        out["K%d" % l], out["Ki%d" % l] = R.get_K(l)
        out["pyr%d" % l] = R.get_pyr(l)
        out["res%d" % l] = R.calcRes(l, start, 0.02, 1.0, 20.0)
        out["warped%d" % l] = R.warped()
        out["H%d" % l], out["b%d" % l] = R.calcGS(l, 0.02, 1.0)
    ok, pose, aff, last_res, flow, _ = R.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
    out.update(track_ok=np.array(ok), track_pose=pose, track_aff=aff, track_lastres=last_res, track_flow=flow)
    return out


BACKEND_CFG = dict(w=200, h=96, nF=4, pts_per_kf=60, seed=7, calib=dict(fx=150., fy=152., cx=99.5, cy=47.5))
TRACKER_CFG = dict(w=160, h=120, levels=3, n_points=200, seed=3, gt_xi=[0.03, -0.02, 0.05, 0.004, -0.006, 0.002], gt_aff=(0.03, 1.5),
                   calib=dict(fx=140.0, fy=145.0, cx=79.3, cy=59.6))


def main():
    L = refpin.ref_lib()
    if L is None:
        sys.exit("oracle/_ref/libref.so is missing: make -C oracle ref (needs /root/reference)")
    out = refpin.run(L, "ref_")
    st = refpin.ref_settings(L)
    out["settings_names"] = np.array(sorted(st))
    out["settings_values"] = np.array([st[k] for k in sorted(st)])
    import ctypes as C  # noqa: E402
    pat = (C.c_int * 80)()
    npat = L.ref_pattern(pat)
    out["pattern"] = np.array(list(pat)[:2 * npat], np.int32).reshape(npat, 2)
    # the SCALE_* macros of src/FullSystem/HessianBlocks.h:33-40 (a header that cannot be compiled here: read as text)
    import re  # noqa: E402
    hb = open("/root/reference/src/FullSystem/HessianBlocks.h").read()
    sc = dict(re.findall(r"#define (SCALE_[A-Z_]+) ([0-9.]+)f", hb))
    out["scale_names"] = np.array(sorted(sc))
    out["scale_values"] = np.array([float(sc[k]) for k in sorted(sc)])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ref_pin.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})

    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
    np.savez_compressed(os.path.join(gdir, "ref_pin_backend.npz"), **backend_reference_outputs(BACKEND_CFG))
    np.savez_compressed(os.path.join(gdir, "ref_pin_tracker.npz"), **tracker_reference_outputs(TRACKER_CFG))
    print("wrote ref_pin_backend.npz, ref_pin_tracker.npz")


if __name__ == "__main__":
    main()
