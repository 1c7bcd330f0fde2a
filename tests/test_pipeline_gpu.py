"""One frame through the device entry points the way FullSystem::trackNewCoarse / traceNewCoarse chain them (FullSystem.cpp:283-553),
with the handles sharing images on the device (no copies): back-end window -> reprojector key-frame images, tracker pyramid ->
reprojector new-frame levels and coarse-depth reference.  Every stage is checked against the CPU oracle fed with host copies."""
import numpy as np
import pytest

from common import rel_err

pytestmark = pytest.mark.gpu

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)


def test_frame_pipeline_shares_device_images(orc):
    from oracle.backend import OracleEF
    from oracle.reproject import OracleReprojector
    from oracle.trace import trace_on
    from sdv_loam_amd import api, backend_api, reproject_api, synthetic as syn
    W = syn.make_window(w=320, h=200, nF=4, pts_per_kf=300, seed=21, calib=CAL)
    L, new = 3, W.nF - 1
    # ---- window on the device (back end) and the new frame on the tracker --------------------------------
    EG = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    TG = api.CoarseTracker(W.w, W.h, L, max_points=W.w * W.h, max_batch=2)
    TO = orc.OracleTracker(W.w, W.h, L)
    ref = new - 1                                           # newest key-frame = tracking reference
    sel = W.host == ref
    tup = (W.u[sel].astype(np.int32), W.v[sel].astype(np.int32), W.idepth[sel], np.full(int(sel.sum()), np.float32(3.0)))
    for T in (TG, TO):
        T.makeK(**CAL)
        T.set_new_image(W.images[ref], 1.0)                 # lastRef->dIp
        T.makeCoarseDepth(*tup)                             # setCoarseTrackingRef
        T.set_ref_frame(1.0, 0.0, 0.0)
        T.set_new_image(W.images[new], 1.0)                 # the frame being tracked
    gt = orc.se3_mul(W.gt_worldToCam[new], orc.se3_inverse(W.gt_worldToCam[ref]))
    start = orc.se3_mul(orc.se3_exp(syn.perturbation(21, 0.02, 0.003)), gt)
    okg, pg, *_ = TG.trackNewestCoarse(start, (0.0, 0.0), L - 1)
    oko, po, *_ = TO.trackNewestCoarse(start, (0.0, 0.0), L - 1)
    assert okg == oko and oko
    d = lambda p: orc.se3_log(orc.se3_mul(p, orc.se3_inverse(start)))   # noqa: E731
    assert rel_err(d(pg), d(po)) < 1e-4
    # ---- Reprojector: key-frame images borrowed from the window, new-frame pyramid borrowed from the tracker ------------
    P = syn.make_reproject_problem(W, levels=L, seed=21, pose_err=(0.0, 0.0))
    cur_c2w = orc.se3_inverse(orc.se3_mul(po, W.gt_worldToCam[ref]))     # camToWorld of the new frame from the tracked pose (oracle's, both sides)
    RG = reproject_api.Reprojector(W.w, W.h, L, max_frames=8, max_points=4096)
    RO = OracleReprojector(W.w, W.h, L)
    RG.set_calib(**CAL)
    RO.set_calib(**CAL)
    for k in range(new):
        RG.set_frame(k, P.frame_poses7[k], None, dev_ptr=EG.frame_image_dev(k))
        RO.set_frame(k, P.frame_poses7[k], P.frame_images[k])
    RG.set_cur(cur_c2w, dev_ptrs=[TG.pyr_dev(l) for l in range(L)])
    RO.set_cur(cur_c2w, P.cur_pyr)
    g = RG.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    px0, cell, q = RO.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, lvl = RO.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    cand = cell >= 0
    assert np.array_equal(g["cell"], cell) and np.array_equal(g["quality"], q)
    assert np.array_equal(g["success"][cand], ok[cand]) and np.array_equal(g["px"][cand & ok], pm[cand & ok])
    assert (cand & ok).sum() > 100
    # ---- structPoseEstimation on the matches -------------------------------------------------------------------------
    m = np.nonzero(cand & ok)[0][:1200]
    a = (P.u[m], P.v[m], P.idepth[m], P.host_idx[m], P.frame_poses7, pm[m])
    sg, tg, _ = TG.structPoseEstimation(cur_c2w, *a)
    so, to, _ = TO.structPoseEstimation(cur_c2w, *a)
    assert len(tg) == len(to) and np.array_equal(tg[:, 4], to[:, 4])
    dd = lambda p: orc.se3_log(orc.se3_mul(orc.se3_inverse(cur_c2w), p))  # noqa: E731
    assert np.linalg.norm(dd(sg) - dd(so)) <= 1e-4 * max(np.linalg.norm(dd(so)), 1e-9) + 1e-12
    # ---- traceNewCoarse: immature points of the key-frames on the tracker's new frame ---------------------------------
    TP = syn.make_trace_problem(W, target=new, seed=21)
    TG.traceSetPoints(TP.u, TP.v, TP.energyTH, TP.gradH, TP.color, TP.weights, TP.host_idx)
    stg = TG.tracePoints(TP.KRKi, TP.Kt, TP.aff, TP.idepth_min, TP.idepth_max, TP.quality, TP.status)
    sto = trace_on(TP, TP.dI, TP.idepth_min, TP.idepth_max, TP.quality, TP.status)
    for k in stg:
        assert np.array_equal(stg[k], sto[k], equal_nan=True), k
    # ---- activation candidates on the window handle, then a few optimize iterations: the window is still consistent ------
    EO = OracleEF(W.w, W.h).load(W)
    good = sto["status"] == 0
    idx = np.nonzero(good)[0][:500]
    hosts = TP.host_idx[idx]
    ia = (hosts, TP.u[idx], TP.v[idx], sto["idepth_min"][idx], sto["idepth_max"][idx], TP.energyTH[idx], TP.color[idx], TP.weights[idx],
          np.zeros(len(idx), np.uint8))
    rg, ro = EG.optimizeImmature(*ia), EO.optimizeImmature(*ia)
    assert np.array_equal(rg[0], ro[0]) and np.array_equal(rg[1], ro[1], equal_nan=True) and np.array_equal(rg[2], ro[2])
    trg, tro = EG.optimize(3), EO.optimize(3)
    assert len(trg) == len(tro) and np.array_equal(trg[:, 2], tro[:, 2])
