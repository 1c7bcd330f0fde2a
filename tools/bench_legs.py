"""Legs of bench.py that time a KEY-FRAME, not a loop body: the window edited in place through the C ABI, and the
reference's own FullSystem::optimize host code with libsdvgn linked in (both drop-in forms) beside the all-CPU reference.  N = 1, rank 0 only."""
import time

import numpy as np


def keyframe_update_leg(steps=8, world=None):
    """value_keyframe_update_inclusive: a sequence of `steps` key-frame insertions + marginalisations on ONE resident window
    (tools/exp_keyframe_update.py), every edit, the commit and optimize(6) + the tail's linearizeAll(true) inside the timed region; next to it
    the same key-frames through the whole-plane setters (value_window_reload_inclusive)."""
    import torch
    from sdv_loam_amd import backend_api as api, synthetic as syn
    from tools import exp_keyframe_update as X
    W = world or X.world()
    RW = X.ResidentWindow(W, list(range(8)))
    RW.G.optimize(6, want_trace=False, fixed_its=True); RW.G.optimize_finish()
    for _ in range(2):
        RW.prepare(); RW.step()
    torch.cuda.synchronize()
    t = []
    for _ in range(steps):
        RW.prepare()
        t0 = time.perf_counter()
        RW.step()
        t.append(time.perf_counter() - t0)
    prof = {}
    for _ in range(4):
        RW.prepare(); RW.step(prof=prof)
    # the same steps from a plain-C host loop on include/sdvgn.h (tools/kf_host_loop.c): the ABI's consumer is a C++ host loop
    tc = None
    try:
        ch = X.CHostLoop(RW)
        ch.run(2)
        tc = ch.run(steps)
    except Exception as ex:  # noqa: BLE001
        c_err = repr(ex)
    n = 4 + 6 * 8
    S = syn.subwindow(W, RW.win, np.concatenate([W.pts_of[f] for f in RW.win]), HM=W.HM[:n, :n], bM=W.bM[:n])
    G2 = api.EnergyFunctional(W.w, W.h, max_points=S.nP + 4096)
    tl = []
    for _ in range(5):
        t0 = time.perf_counter()
        G2.load(S, raw_images=True); G2.optimize(6, want_trace=False, fixed_its=True); G2.optimize_finish()
        tl.append(time.perf_counter() - t0)
    t, tl = np.array(t), np.array(tl[1:])
    tm = tc if tc is not None else t
    return dict(value=float(6 / np.median(tm)), unit="GN iters/s", steps=steps, bodies_per_keyframe=6,
                host_loop="C (tools/kf_host_loop.c on include/sdvgn.h)" if tc is not None else "Python (ctypes): the C harness did not load (%s)" % c_err,
                ms_per_keyframe=dict(median=float(1e3 * np.median(tm)), min=float(1e3 * tm.min()), max=float(1e3 * tm.max())),
                value_python_host_loop=float(6 / np.median(t)), ms_per_keyframe_python_host_loop=float(1e3 * np.median(t)),
                host_phases_us={k: float(1e6 * np.median(v)) for k, v in prof.items()},
                value_window_reload_inclusive=float(6 / np.median(tl)), ms_per_keyframe_reload=float(1e3 * np.median(tl)),
                note="per key-frame: removePoint x 2000, marginalizeFrame, insertFrame (1.87 MB raw image from pinned memory, level 0 built on the device), "
                     "insertPoint x 2000, insertResidual x 28000, makeIDX (device-side re-pack), setAdjointsF, setPrecalcValues, optimize(6 bodies), "
                     "linearizeAll(true); bit-identical with a reload of the same graph (tests/test_window_update_gpu.py).  "
                     "value_window_reload_inclusive: the same key-frame through sdvgn_ef_set_frames / _set_frame_image_raw x 8 / _set_points / _set_residuals")


def dropin_legs(W9, want_cpu=True):
    """The reference's OWN host loop (oracle/ref_glue_ef.cpp drives FullSystem::optimize of the unmodified reference objects), three ways, on the same
    two key-frames of a 9-frame world at the named shape: all-CPU (libref.so), form A (EnergyFunctional::solveSystemF on the GPU, linearize + loop on
    the host: libref_dropin.so), form B (FullSystem::optimize = sdvgn_ef_optimize on a resident window: libref_dropin_opt.so).  Timed: the
    FullSystem::optimize call alone, exactly 6 loop bodies (setting_minOptIterations = 6), wall clock inside the glue -- for form B that includes the
    graph walk, the edits, the commit, the device work and the write-back into the reference's objects."""
    from oracle import dropin, refpin
    from oracle.backend import RefEF
    from sdv_loam_amd import synthetic as syn
    if refpin.ref_lib() is None or dropin.dropin_lib() is None or dropin.dropin_opt_lib() is None:
        return dict(error="oracle/_ref/libref*.so not present on this machine")
    nF0 = W9.nF - 1
    frames = list(range(nF0))
    big = np.nonzero(np.isin(W9.host, frames))[0]
    n0 = 4 + 6 * nF0
    S = syn.subwindow(W9, frames, big, HM=W9.HM[:n0, :n0], bM=W9.bM[:n0])
    rof = (W9.r_point.astype(np.int64) * W9.nF + W9.r_target)
    order = np.argsort(rof)

    def rrow(p, t):       # index of W9's residual (point p, target t)
        return order[np.searchsorted(rof[order], np.asarray(p, np.int64) * W9.nF + t)]

    out = {}
    import os
    legs = [("dropin_optimize", dropin.DropinOptEF), ("dropin_optimize_4_host_threads", dropin.DropinOptEF), ("dropin_solveSystemF", dropin.DropinEF)]
    if want_cpu:
        legs.append(("cpu_reference", RefEF))
    for key, cls in legs:
        if key == "dropin_optimize_4_host_threads":
            os.environ["SDVGN_DROPIN_THREADS"] = "4"
        else:
            os.environ.pop("SDVGN_DROPIN_THREADS", None)
        E = cls(S.w, S.h).set_levels(3).load(S)
        E.compute_nullspaces()
        E.keyframe_tail(6, [1], min_its=6)                                         # key-frame 1: optimize (first call: everything is new to the GPU), frame 1 marginalised
        t_first = E.last_optimize_seconds
        new_big = W9.nF - 1
        win_big = [f for f in frames if f != 1] + [new_big]
        hosts = E.point_hosts()
        k = E.append_frame(W9.evalPT[new_big], W9.state[new_big], W9.state_zero[new_big], int(W9.frameID[new_big]), 1.0, W9.frameEnergyTH[new_big], W9.pyr0[new_big])
        old = np.nonzero(hosts >= 0)[0]
        rr = rrow(big[old], new_big)
        E.append_residuals(old, np.full(len(old), k), W9.r_hasMatcher[rr], W9.r_matcher[rr])
        E.setAdjointsF(); E.setPrecalcValues()
        _, steps, _, _ = E.optimize_full(6, min_its=6)                             # key-frame 2: the steady state
        t_kf = E.last_seconds
        row = dict(its_per_s=6 / t_kf, ms_per_optimize_call=1e3 * t_kf, first_call_its_per_s=6 / t_first, bodies=len(steps), points=int((hosts >= 0).sum()),
                   residuals_inserted_by_the_keyframe=int(len(old)))
        if hasattr(E, "gpu_stats"):
            row["gpu_window"] = E.gpu_stats()
            row["host_threads_of_the_graph_walks"] = int(os.environ.get("SDVGN_DROPIN_THREADS", "1"))
        out[key] = row
        del E
    out["note"] = ("FullSystem::optimize of the reference's own host loop on the second of two key-frames (8 key-frames in the window, ~16 000 points; between "
                   "them the reference's removeOutliers / flagPointsForRemoval / marginalizePointsF / marginalizeFrame / insertFrame / insertResidual ran on the "
                   "host): its_per_s = 6 bodies / wall time of the call.  dropin_optimize: oracle/dropin/FullSystemOptimizeGPU.cpp (resident window, one image "
                   "upload per key-frame, graph walk + edits + write-back included; dropin_optimize_4_host_threads: the two walks over the reference's heap "
                   "objects on 4 host threads -- no faster, the serial part dominates); dropin_solveSystemF: oracle/dropin/EnergyFunctionalGPU.cpp (every plane "
                   "and the CPU-computed Jacobians re-sent per solve); cpu_reference: libref.so, 1 thread, Eigen stand-in (a lower bound of a real Eigen build)")
    return out


def frame_legs(want_cpu=True):
    """The reference's per-FRAME and per-key-frame host code around one window at the named shape (1241x376, 8 key-frames), all-CPU (libref.so) against
    form B+ (libref_dropin_frame.so: traceNewCoarse, the activation's optimizeImmaturePoint batch, optimize and the next tracking template on the GPU, bound at
    the reference's own call sites -- oracle/dropin/FullSystemFrameGPU.cpp).  Timed inside the glue, the reference's call alone: FullSystem::traceNewCoarse
    on two new frames (~6 400 immature points), FullSystem::activatePointsMT, then makeKeyFrame's tail (optimize .. marginalizeFrame, setCoarseTrackingRef)."""
    import oracle as orc
    from oracle import dropin, refpin
    from oracle.backend import RefEF
    from sdv_loam_amd import synthetic as syn
    if refpin.ref_lib() is None or dropin.dropin_frame_lib() is None:
        return dict(error="oracle/_ref/libref.so / libref_dropin_frame.so not present on this machine")
    WB = syn.make_window(w=1241, h=376, nF=10, pts_per_kf=2000, seed=1, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02, spacing=0.5)
    nW = WB.nF - 2
    frames = list(range(nW))
    rng = np.random.default_rng(3)
    in_win = np.isin(WB.host, frames)
    active = np.nonzero(in_win & (rng.random(WB.nP) < 0.6))[0]
    imm = np.nonzero(in_win & ~np.isin(np.arange(WB.nP), active))[0]
    iu, iv = np.round(WB.u[imm]).astype(np.int32), np.round(WB.v[imm]).astype(np.int32)
    ok = (iu > 8) & (iu < WB.w - 8) & (iv > 8) & (iv < WB.h - 8)
    imm, iu, iv = imm[ok], iu[ok], iv[ok]
    n0 = 4 + 6 * nW
    S = syn.subwindow(WB, frames, active, HM=WB.HM[:n0, :n0], bM=WB.bM[:n0])
    out = {}
    legs = [("dropin_frame", dropin.DropinFrameEF)] + ([("cpu_reference", RefEF)] if want_cpu else [])
    for key, cls in legs:
        E = cls(S.w, S.h).set_levels(3).load(S)
        E.compute_nullspaces()
        kept = E.add_immature(WB.host[imm], iu, iv, np.zeros(len(imm), np.float32), np.full(len(imm), np.nan, np.float32))
        tr = []
        for f in (WB.nF - 2, WB.nF - 1, WB.nF - 2, WB.nF - 1):          # (two more passes over the same two frames: the steady state of the GPU side)
            E.trace_new_frame(WB.images[f], orc.se3_inverse(WB.gt_worldToCam[f]))
            tr.append(E.last_seconds)
        n_act = E.activate_points()[0]
        t_act = E.last_seconds
        E.setAdjointsF(); E.setPrecalcValues()
        E.keyframe_tail(6, [0], min_its=6)
        row = dict(immature_points=int(kept), ms_trace_new_coarse_first=1e3 * tr[0], ms_trace_new_coarse=1e3 * float(np.median(tr[1:])), points_activated=int(n_act),
                   ms_activate_points=1e3 * t_act, ms_keyframe_tail=1e3 * E.last_seconds, ms_optimize_in_tail=1e3 * E.last_optimize_seconds)
        row["ms_per_frame"] = row["ms_trace_new_coarse"]
        row["ms_per_keyframe"] = row["ms_trace_new_coarse"] + row["ms_activate_points"] + row["ms_keyframe_tail"]
        if hasattr(E, "frame_stats"):
            row["gpu_call_sites"] = E.frame_stats()
        out[key] = row
        del E
    out["note"] = ("the reference's own FullSystem::traceNewCoarse / activatePointsMT / makeKeyFrame tail on a window of 8 key-frames at the named shape; dropin_frame = "
                   "libref_dropin_frame.so (form B+); ms_per_frame = traceNewCoarse, ms_per_keyframe = traceNewCoarse + activatePointsMT + optimize .. marginalizeFrame "
                   "+ setCoarseTrackingRef (tracking itself -- trackNewCoarse with reprojectMap and structPoseEstimation -- is timed in the tracker extras)")
    return out
