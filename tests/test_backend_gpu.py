"""Parity of the HIP sliding-window back end (through the C ABI) against the CPU oracle on the same seeded window.

Tolerances: residual states / activity flags exact; Jacobians and per-residual energies bit-exact (same float32
operations, no FMA contraction); (host,target) accumulators rel 1e-5; H, b of the stitched system rel 1e-5 / 1e-4;
solution x and point steps rel 1e-4 (BASELINE.json north_star: 1e-4 relative on increments)."""
import copy
import os

import numpy as np
import pytest

from common import rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(sdvgn_lib):
    from sdv_loam_amd import backend_api
    return backend_api


@pytest.fixture(scope="module")
def window():
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=2, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))


def low_thresholds(W):
    """Older frames get outlier thresholds below the newest frame's, so that the threshold setNewFrameEnergyTH recomputes after every
    linearizeAll (max(host TH, target TH), Residuals.cpp:212) actually decides IN / OUTLIER for the residuals that involve the newest frame."""
    W = copy.copy(W)
    W.frameEnergyTH = np.concatenate([np.linspace(150, 200, W.nF - 1), [300]]).astype(np.float32)
    return W


def pair(api, orc, W, **kw):
    from oracle.backend import OracleEF
    G = api.EnergyFunctional(W.w, W.h, max_points=max(W.nP, 16)).load(W, **kw)
    O = OracleEF(W.w, W.h).load(W)
    return G, O


def check_linearize(G, O):
    eg = G.linearizeAll()
    eo = O.linearizeAll()
    sg, so = G.residual_state(), O.residual_state()
    assert np.array_equal(sg["new_state"], so["new_state"])
    assert np.array_equal(sg["new_energy"], so["new_energy"].astype(np.float32))
    assert np.array_equal(sg["energy_with_outlier"], so["energy_with_outlier"].astype(np.float32))
    Jg, Jo = G.residual_J(0), O.residual_J(0)
    touched = so["new_state"] != 1           # OOB residuals leave J untouched
    assert np.array_equal(Jg[touched].view(np.uint32), Jo[touched].view(np.uint32))
    assert rel_err(eg, eo) < 1e-6
    assert np.array_equal(G.frame_energy_th(), O.frame_energy_th())          # setNewFrameEnergyTH at the end of linearizeAll: exact quantile
    return sg, so


def check_solve(G, O, iteration, lam):
    xg = G.solveSystemF(iteration, lam)
    O.solveSystemF(iteration, lam)
    so = O.system()
    sg = G.system()
    accg, nres = G.top_acc()
    acco = O.top_acc()
    assert nres == O.resInA()
    for k in range(accg.shape[0]):
        ref = acco[k][np.ix_(list(range(10)) + [12], list(range(10)) + [12])].astype(np.float64)
        if np.abs(ref).max() > 0:
            assert rel_err(accg[k], ref) < 1e-5, k
        else:
            assert np.all(accg[k] == 0)
    pg, po = G.points(), O.points()
    assert rel_err(pg[:, 0], po[:, 0]) < 1e-6 and rel_err(pg[:, 1], po[:, 1]) < 1e-5 and rel_err(pg[:, 2:6], po[:, 2:6]) < 1e-5
    assert rel_err(pg[:, 6], po[:, 6]) < 1e-6 and rel_err(pg[:, 7], po[:, 7]) < 1e-5
    assert rel_err(sg["HA"], so["HA"]) < 1e-5 and rel_err(sg["bA"], so["bA"]) < 1e-5
    assert rel_err(sg["Hsc"], so["Hsc"]) < 1e-5 and rel_err(sg["bsc"], so["bsc"]) < 1e-4
    assert rel_err(sg["HFinal"], so["HFinal"]) < 1e-5 and rel_err(sg["bFinal"], so["bFinal"]) < 1e-4
    assert rel_err(xg, so["x"]) < 1e-4
    assert rel_err(pg[:, 8], po[:, 8]) < 1e-4
    return xg


def test_linearize_apply_solve_parity(api, orc, window):
    G, O = pair(api, orc, window)
    sg, so = check_linearize(G, O)
    G.applyRes(); O.applyRes()
    s2g, s2o = G.residual_state(), O.residual_state()
    assert np.array_equal(s2g["state"], s2o["state"]) and np.array_equal(s2g["active"], s2o["active"])
    assert np.array_equal(G.residual_J(1).view(np.uint32)[s2o["active"] == 1], O.residual_J(1).view(np.uint32)[s2o["active"] == 1])
    check_solve(G, O, 0, 0.1)
    sg = G.system()
    # device stitch (k_ef_stitch: HFinal, bFinal) vs the host stitch of the same accumulators (HA, Hsc): fp64 round-off only
    assert rel_err(sg["HFinal"], sg["HA"] + window.HM - sg["Hsc"]) < 1e-11
    d = np.concatenate([window.value_minus_value_zero.astype(np.float32).astype(np.float64), (window.state - window.state_zero)[:, :6].reshape(-1)])
    assert rel_err(sg["bFinal"], sg["bA"] + window.bM + window.HM @ d - sg["bsc"]) < 1e-10
    check_solve(G, O, 1, 1e-3)
    # second linearisation at the same state: new J goes to the other buffer, EF-side J unchanged, deterministic
    check_linearize(G, O)
    x1 = G.solveSystemF(0, 0.1)
    x2 = G.solveSystemF(0, 0.1)
    assert np.array_equal(x1, x2)


def test_images_built_on_device(api, orc, window):
    G, O = pair(api, orc, window, raw_images=True)
    check_linearize(G, O)


def test_nullspace_projection_iteration2(api, orc, window):
    W = copy.copy(window)
    rng = np.random.default_rng(5)
    W.nullspaces = rng.normal(size=(7, 4 + 6 * W.nF))
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    x0 = check_solve(G, O, 0, 0.1)
    x2 = check_solve(G, O, 2, 0.1)
    assert rel_err(x2, x0) > 1e-3                     # the projection did something
    Q, _ = np.linalg.qr(W.nullspaces.T)
    assert np.abs(Q.T @ x2).max() < 1e-9 * np.abs(x2).max() + 1e-12


def test_edge_cases(api, orc, window):
    """OOB / matcher-less residuals, inactive points (no good residual), all-sensor points, ragged hosts (one frame hosts
    no point), residuals already linearised (addPoint<1> path)."""
    W = copy.copy(window)
    W.r_hasMatcher = W.r_hasMatcher.copy(); W.r_hasMatcher[::5] = 0
    W.r_state = W.r_state.copy(); W.r_state[2::9] = 1
    W.isFromSensor = W.isFromSensor.copy(); W.isFromSensor[:50] = 1
    # remove every point hosted by frame 2 (ragged): keep arrays consistent
    keep_p = W.host != 2
    remap = -np.ones(W.nP, int); remap[keep_p] = np.arange(keep_p.sum())
    keep_r = keep_p[W.r_point]
    for name in ("host", "u", "v", "idepth", "idepth_zero", "color", "weights", "hasDepthPrior", "isFromSensor"):
        setattr(W, name, getattr(W, name)[keep_p])
    W.r_point = remap[W.r_point[keep_r]].astype(np.int32)
    for name in ("r_target", "r_matcher", "r_state", "r_hasMatcher", "r_isLinearized", "r_isActive"):
        setattr(W, name, getattr(W, name)[keep_r])
    W.nP, W.nR = int(keep_p.sum()), int(keep_r.sum())
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    check_solve(G, O, 0, 0.1)
    st = O.residual_state()
    assert np.all(st["state"][W.r_hasMatcher == 0] == 1)


def test_full_size_cfg3(api, orc):
    """BASELINE.json configs[2]: KITTI-00 calib, 1241x376, 8 key-frames x 2000 points, 112 000 residuals."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)
    assert W.nR == 112000
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    x = check_solve(G, O, 0, 0.1)
    # size-independent property: H x = b for the damped system
    s = G.system()
    n = G.dim
    Hd = s["HFinal"].copy(); Hd[np.arange(n), np.arange(n)] *= 1.1
    assert rel_err(Hd @ x, s["bFinal"]) < 1e-6


def test_full_size_shard_linearity(api):
    """Size-independent property at BASELINE.json's full size (configs[2]/[3]): the packed accumulator buffer is a sum over residuals
    and a host-frame shard only touches its own (host, target) tiles and host Schur tiles, so the buffers of disjoint shards add up
    to the unsharded buffer EXACTLY (every entry has exactly one non-zero contributor; resInA adds as integers).  Shardings 8 = 4+4,
    3+5 and 8 x 1 (the one-key-frame-per-GPU layout of configs[3])."""
    import torch
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00)

    def fetch(h0, h1):
        G = api.EnergyFunctional(W.w, W.h, max_points=W.nP)
        acc = torch.zeros(8 * 8 * 121 + 8 * 1431 + 1, dtype=torch.float64, device="cuda")      # capacity for nF = 8
        stats = torch.zeros(4 + W.nP, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        G._check(G.L.sdvgn_ef_set_external_buffers(G.h_, acc.data_ptr(), acc.numel(), stats.data_ptr(), stats.numel()))
        G.set_host_range(h0, h1)
        G.load(W)
        G.linearizeAll(); G.applyRes()
        G.accumulate()                                  # the packed buffer lands in `acc` (library stream)
        torch.cuda.synchronize()
        out = acc.cpu().numpy().copy()
        G.close()
        return out

    full = fetch(0, 8)
    assert full[-1] > 50000 and np.abs(full).max() > 0                         # resInA: active residuals of the window
    for cuts in ((0, 4, 8), (0, 3, 8), tuple(range(9))):
        parts = [fetch(a, b) for a, b in zip(cuts[:-1], cuts[1:])]
        assert np.array_equal(sum(parts), full), cuts
        nz = sum((p != 0).astype(int) for p in parts)
        assert nz.max() <= 1 or np.array_equal(np.flatnonzero(nz > 1), [len(full) - 1])   # only resInA is shared


def check_optimize(G, O, its=6, **kw):
    tg = G.optimize(its, **kw)
    to = O.optimize(its, **{k: v for k, v in kw.items() if k == "fixed_its"})
    n = G.dim
    assert len(tg) == len(to) and len(to) >= 1
    assert np.array_equal(tg[:, [0, 1, 2, 6]], to[:, [0, 1, 2, 6]])          # iteration, lambda, accepted, canbreak
    # energies: the accept test compares the SUM E + E_L + E_M; the two small terms are checked against its scale
    assert np.allclose(tg[:, 3], to[:, 3], rtol=1e-5) and np.allclose(tg[:, 4:6], to[:, 4:6], rtol=1e-4, atol=1e-9 * np.abs(to[:, 3]).max())
    for i in range(len(to)):
        assert rel_err(tg[i, 7:7 + n], to[i, 7:7 + n]) < 1e-4                 # increments x of every iteration
    # the newest frame's outlier threshold after every trial linearizeAll.  The quantile selection itself is exact (bit-identical on
    # identical energies: check_linearize); inside the loop the trial states already differ by the solve's rounding (x rel 1e-4)
    assert np.allclose(tg[:, 7 + n], to[:, 7 + n], rtol=1e-4)
    assert np.allclose(G.frame_energy_th(), O.frame_energy_th(), rtol=1e-4)
    vg, sg, ig = G.state()
    vo, so, io = O.state()
    assert np.allclose(vg, vo, rtol=1e-9) and rel_err(sg, so) < 1e-4
    assert rel_err(ig, io) < 1e-6
    rg, ro = G.residual_state(), O.residual_state()
    assert np.array_equal(rg["state"], ro["state"]) and np.array_equal(rg["active"], ro["active"])
    return tg, to


@pytest.mark.parametrize("seed", [2, 3])
@pytest.mark.parametrize("low_th", [False, True])
def test_optimize_loop_parity(api, orc, seed, low_th):
    """Whole FullSystem::optimize loop (b8): same accept/reject sequence, lambda schedule, x per iteration, per-iteration outlier
    threshold of the newest frame (setNewFrameEnergyTH inside every linearizeAll) and final state."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    if low_th:
        W = low_thresholds(W)
    G, O = pair(api, orc, W)
    tg, to = check_optimize(G, O)
    if low_th:
        assert len(set(to[:, -1])) > 1 or len(to) == 1                        # the threshold moved between the trial linearisations


@pytest.mark.parametrize("nF,pts", [(8, 2000), (7, 2000), (7, 286), (6, 700)])
def test_optimize_loop_parity_full_size(api, orc, nF, pts):
    """The same loop parity at BASELINE.json configs[2] size (8 x 2000, 112 000 residuals), at the reference's own maximum window
    (setting_maxFrames = 7) with 2000 points per key-frame and with ~2000 points in the WHOLE window (setting_desiredPointDensity,
    settings.cpp:46-47,52-53), and at nF = 6; thresholds that matter, perturbed far enough that steps are accepted and rejected."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=nF, pts_per_kf=pts, seed=nF, calib=syn.KITTI00, state_sigma=1e-3, idepth_sigma=0.01)
    W = low_thresholds(W)
    G, O = pair(api, orc, W)
    tg, to = check_optimize(G, O, 6)
    e, rb, ng, rm = G.optimize_finish()
    eo, rbo, ngo, rmo = O.optimize_finish()
    assert rel_err(e, eo) < 1e-5 and np.array_equal(rm, rmo) and np.array_equal(ng, ngo) and np.allclose(rb, rbo, rtol=1e-4, atol=1e-6)
    assert np.allclose(G.frame_energy_th(), O.frame_energy_th(), rtol=1e-4)


@pytest.mark.parametrize("seed", [2, 3])
def test_optimize_finish_parity(api, orc, seed):
    """Tail of FullSystem::optimize (FullSystemOptimize.cpp:460-470): setEvalPT on the newest frame, adjoints, precalc,
    linearizeAll(true) -- energies, states, dropped residuals, per-point bookkeeping and the final threshold against the oracle; the
    window then solves on (new linearisation point, residuals gone) like the oracle's."""
    from sdv_loam_amd import synthetic as syn
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)))
    W.r_hasMatcher = W.r_hasMatcher.copy(); W.r_hasMatcher[::13] = 0           # some residuals go OOB -> toRemove
    G, O = pair(api, orc, W)
    # on the freshly loaded window (identical inputs) everything the tail computes is bit-identical ...
    G.resetOOB(); O.resetOOB()
    e, rb, ng, rm = G.optimize_finish()
    eo, rbo, ngo, rmo = O.optimize_finish()
    assert rm.sum() > 0 and np.array_equal(rm, rmo)
    assert rel_err(e, eo) < 1e-6
    assert np.array_equal(ng, ngo) and np.array_equal(rb, rbo)                 # relBS: same float operations
    assert np.array_equal(G.frame_energy_th(), O.frame_energy_th())
    rg, ro = G.residual_state(), O.residual_state()
    assert np.array_equal(rg["new_energy"], ro["new_energy"].astype(np.float32))
    # ... and after a whole optimize (states equal to 1e-4 only) it follows the oracle within the tolerance
    G, O = pair(api, orc, W)
    check_optimize(G, O)
    e, rb, ng, rm = G.optimize_finish()
    eo, rbo, ngo, rmo = O.optimize_finish()
    assert rm.sum() > 0 and np.array_equal(rm, rmo)
    assert rel_err(e, eo) < 1e-5
    assert np.array_equal(ng, ngo) and np.allclose(rb, rbo, rtol=1e-4, atol=1e-6)
    assert np.allclose(G.frame_energy_th(), O.frame_energy_th(), rtol=1e-4)
    sg, so = G.state()[1], O.state()[1]
    assert np.all(sg[-1, :6] == 0) and np.array_equal(sg[-1, 6:8], so[-1, 6:8])
    rg, ro = G.residual_state(), O.residual_state()
    keep = rmo == 0
    assert np.array_equal(rg["state"][keep], ro["state"][keep]) and np.array_equal(rg["active"][keep], ro["active"][keep])
    assert not rg["active"][~keep].any()
    xg = G.solveSystemF(0, 0.1)                                                # the window goes on without the dropped residuals
    O.solveSystemF(0, 0.1)
    assert rel_err(xg, O.system()["x"]) < 1e-4


def test_handle_reuse_growing_window(api, orc):
    """One handle, windows of 4 then 8 then 5 key-frames (the reference's window grows 2 -> 7 and then cycles): the host-frame range
    follows nF, and a new frame set invalidates the point / residual tables until they are set again."""
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    cal = dict(fx=400., fy=410., cx=319.5, cy=119.5)
    Ws = [syn.make_window(w=640, h=240, nF=nF, pts_per_kf=200, seed=10 + nF, calib=cal) for nF in (4, 8, 5)]
    G = api.EnergyFunctional(640, 240, max_points=max(W.nP for W in Ws))
    for W in Ws:
        G.load(W)
        O = OracleEF(W.w, W.h).load(W)
        check_linearize(G, O)
        G.applyRes(); O.applyRes()
        check_solve(G, O, 0, 0.1)
    # frames changed, points not yet set again: refuse instead of reading the stale layout
    W = Ws[1]
    c = np.ascontiguousarray
    G._check(G.L.sdvgn_ef_set_frames(G.h_, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1),
                                     c(W.state_zero, np.float64).reshape(-1), c(W.frameID, np.int32), c(W.ab_exposure, np.float32),
                                     c(W.frameEnergyTH, np.float32)))
    G.setAdjointsF(); G.setPrecalcValues()
    import ctypes as C
    assert G.L.sdvgn_ef_optimize(G.h_, 2, 0, None, 0, 0) < 0
    assert G.L.sdvgn_ef_solve_system(G.h_, 0, C.c_double(0.1), None) < 0
    assert G.L.sdvgn_ef_linearize_all(G.h_, None) < 0


@pytest.mark.parametrize("seed", [3, 5])
def test_first_step_rejected_with_idepth_zero_offset(api, orc, seed):
    """Window loaded with idepth != idepth_zero whose very first step is rejected: the reference's loadSateBackup then also moves
    idepth_zero to the backed-up idepth (FullSystemOptimize.cpp:276-277) -- the one restore the pointer swap alone does not
    reproduce.  Trace, final state and the next solve must follow the oracle, and the three loop variants must stay bit-identical."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    W.idepth_zero = (W.idepth + np.random.default_rng(1).normal(0, 2e-4, W.nP)).astype(np.float32)
    G, O = pair(api, orc, W)
    tg, to = G.optimize(6), O.optimize(6)
    assert to[0, 2] == 0                                                      # first step rejected
    assert len(tg) == len(to) and np.array_equal(tg[:, [0, 1, 2, 6]], to[:, [0, 1, 2, 6]])
    assert np.allclose(tg[:, 3:6], to[:, 3:6], rtol=1e-5, atol=1e-6)
    for i in range(len(to)):
        assert rel_err(tg[i, 7:], to[i, 7:]) < 1e-4
    assert rel_err(G.state()[2], O.state()[2]) < 1e-6
    xg = G.solveSystemF(3, 0.1)                                               # idepth_zero / deltaF after the loop feed this solve
    O.solveSystemF(3, 0.1)
    assert rel_err(xg, O.system()["x"]) < 1e-4
    ref = None
    for kw in (dict(), dict(relinearize_on_reject=True), dict(reuse_after_reject=True)):
        B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        out = (B.optimize(6, **kw), B.state()[2], B.points(), B.solveSystemF(3, 0.1))
        if ref is None:
            ref = out
        else:
            for a, b in zip(ref, out):
                assert np.array_equal(a, b)


@pytest.mark.parametrize("seed", [2, 3, 4])
def test_kept_state_equals_relinearize_on_reject(api, orc, seed):
    """A rejected step switches back to the kept state_New* set instead of re-linearising (FullSystemOptimize.cpp:446-449): both
    variants must be bit-identical in every traced quantity, in the final state and in the per-residual state_New* planes."""
    from sdv_loam_amd import synthetic as syn
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)))
    A = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    ta = A.optimize(8, fixed_its=True)
    tb = B.optimize(8, fixed_its=True, relinearize_on_reject=True)
    assert (ta[:, 2] == 0).any()                                              # the reject branch was taken
    assert np.array_equal(ta, tb)
    for x, y in zip(A.state(), B.state()):
        assert np.array_equal(x, y)
    sa, sb = A.residual_state(), B.residual_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    xa, xb = A.solveSystemF(3, 0.1), B.solveSystemF(3, 0.1)                   # and the next solve sees the same system
    assert np.array_equal(xa, xb)


@pytest.mark.parametrize("seed", [2, 5])
def test_untraced_call_counters_and_thresholds(api, orc, seed):
    """A call without a trace (what bench.py times): same final state, thresholds and per-residual planes as the traced call and as the
    literal variant, whatever the loop ended on (the deferred threshold select / re-classification are flushed when it ends), and the
    handle's accepted-step counter equals the trace's; make_resident is idempotent."""
    from sdv_loam_amd import synthetic as syn
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)))
    for its in (1, 2, 3, 5, 7):                                                # ends on accepted and on rejected steps
        A = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        C = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        A.make_resident(); A.make_resident()
        ta = A.optimize(its, fixed_its=True, want_trace=False)
        tb = B.optimize(its, fixed_its=True)
        tc = C.optimize(its, fixed_its=True, relinearize_on_reject=True)
        assert len(ta) == its and A.accepted_steps() == int(tb[:, 2].sum()) == B.accepted_steps() == C.accepted_steps()
        assert np.array_equal(tb, tc)
        for x, y, z in zip(A.state(), B.state(), C.state()):
            assert np.array_equal(x, y) and np.array_equal(y, z)
        assert np.array_equal(A.frame_energy_th(), B.frame_energy_th()) and np.array_equal(B.frame_energy_th(), C.frame_energy_th())
        sa, sb, sc = A.residual_state(), B.residual_state(), C.residual_state()
        for k in sa:
            assert np.array_equal(sa[k], sb[k]) and np.array_equal(sb[k], sc[k]), (its, k)


@pytest.mark.parametrize("seed", [2, 3, 4])
def test_reuse_after_reject_is_bit_identical(api, orc, seed):
    """flags bit2: the body after a rejected step re-uses the stitched system of its predecessor (same state, only lambda changed)
    instead of accumulating and stitching again -- trace, final state and per-residual planes must not change by a bit, also with
    several rejected steps in a row and with an accepted step in between."""
    from sdv_loam_amd import synthetic as syn
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)))
    A = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    ta = A.optimize(12, fixed_its=True)
    tb = B.optimize(12, fixed_its=True, reuse_after_reject=True)
    rej = ta[:, 2] == 0
    assert rej.any() and (rej[:-1] & rej[1:]).any()                           # consecutive rejected steps: the reuse path ran
    assert np.array_equal(ta, tb)
    for x, y in zip(A.state(), B.state()):
        assert np.array_equal(x, y)
    sa, sb = A.residual_state(), B.residual_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(A.points(), B.points())
    xa, xb = A.solveSystemF(3, 0.1), B.solveSystemF(3, 0.1)
    assert np.array_equal(xa, xb)


@pytest.mark.parametrize("seed,its", [(2, 12), (3, 12), (4, 7), (2, 1), (3, 2)])
def test_rejected_case_solved_ahead_is_bit_identical(api, orc, seed, its):
    """The default loop solves the rejected case of every body ahead on the side stream (ef_launch_spec_solve) and, after a rejection, starts
    the next body from that solution instead of accumulating, stitching and factoring again.  Trace, final state, per-residual planes, point
    planes and the next solve must not differ by a bit from the loop that does not (flags bit4) -- with single and consecutive rejections,
    a rejection in the last body, and an accepted step in between."""
    from sdv_loam_amd import synthetic as syn
    kw = {} if seed == 2 else dict(state_sigma=1e-3, idepth_sigma=0.01)
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5), **kw))
    A = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    ta = A.optimize(its, fixed_its=True)
    tb = B.optimize(its, fixed_its=True, no_spec_solve=True)
    rej = ta[:, 2] == 0
    if seed == 2 and its >= 7:
        assert rej.any() and (~rej).any()                                        # rejected and accepted steps
    if seed == 2 and its >= 12:
        assert (rej[:-1] & rej[1:]).any()                                        # consecutive rejections: a speculative solve on top of a speculative solution
    assert np.array_equal(ta, tb)
    for x, y in zip(A.state(), B.state()):
        assert np.array_equal(x, y)
    sa, sb = A.residual_state(), B.residual_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(A.frame_energy_th(), B.frame_energy_th())
    xa, xb = A.solveSystemF(3, 0.1), B.solveSystemF(3, 0.1)
    assert np.array_equal(xa, xb) and np.array_equal(A.points(), B.points())
    # a second call on the same handles (a side-stream solve of the first call may have been left unused)
    assert np.array_equal(A.optimize(4, fixed_its=True), B.optimize(4, fixed_its=True, no_spec_solve=True))


@pytest.mark.parametrize("direct", [True, False])
def test_sharded_path_single_rank_nccl(api, orc, window, direct, monkeypatch):
    """cfg4 plumbing on one GPU: external torch buffers, torch stream, and the collectives of a 1-rank RCCL group -- issued either by
    the library itself (ncclAllReduce resolved from the librccl.so torch loaded; direct=True) or through the torch.distributed
    callback (direct=False) -- must give exactly the single-GPU result."""
    import torch
    import torch.distributed as dist
    from sdv_loam_amd.parallel import ShardedEnergyFunctional
    if not direct:
        monkeypatch.setenv("SDVGN_NO_DIRECT_RCCL", "1")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29733 + int(direct)), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        S = ShardedEnergyFunctional(window, 0, 1, 0, force_collective=True)
        assert S.direct_rccl == direct
        ts = S.optimize(6, want_trace=True)
        if not direct:
            assert S.n_allreduce == len(ts) + 1                 # ONE all-reduce per loop body + one per call (speculative accumulate)
        else:
            assert S.ef.L.sdvgn_ef_collective_count(S.ef.h_) == len(ts) + 1
        G = api.EnergyFunctional(window.w, window.h, max_points=window.nP).load(window)
        tg = G.optimize(6)
        if os.environ.get("SDVGN_DUMP_TRACES"):   # fault hunts (tools/hunt_uaf.sh): both traces to disk, whichever way the comparison goes
            np.savez(os.path.join(os.environ["SDVGN_DUMP_TRACES"], "sharded_trace_direct%d.npz" % int(direct)), ts=np.asarray(ts), tg=np.asarray(tg))
        assert np.array_equal(ts, tg)
        assert np.array_equal(S.ef.state()[2], G.state()[2])
        if direct:
            # a second sharded window of the same process group shares the first one's communicator (reference-counted in the library:
            # no second ncclCommInitRank) and outlives it
            from sdv_loam_amd import parallel
            ident = next(i for g, i in parallel._GROUP_IDS if g is S.group)
            S2 = ShardedEnergyFunctional(window, 0, 1, 0, force_collective=True)
            assert S2.direct_rccl and next(i for g, i in parallel._GROUP_IDS if g is S2.group) == ident
            del S
            assert S2.ef.L.sdvgn_rccl_comm_alive(ident) == 1
            assert np.array_equal(S2.optimize(6, want_trace=True), tg)
            del S2
            import gc
            gc.collect()
            assert G.L.sdvgn_rccl_comm_alive(ident) == 0
        else:
            del S
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("direct", [True, False])
def test_sharded_idepth_zero_offset_first_step_rejected(api, orc, direct, monkeypatch):
    """The sharded loop (ONE collective per body, speculative apply + accumulate) on a window loaded with idepth != idepth_zero whose first
    step is rejected: that body runs the two-collective way and rebuilds the current message after the restore moved idepth_zero
    (backend.hip, `onecoll && !spec`), the following bodies the speculative way.  Trace, states, points and the next solve must be those
    of the plain handle, bit for bit -- through the library's own ncclAllReduce and through the torch.distributed callback on torch-owned
    buffers (this edge was only tested single-GPU)."""
    import torch
    import torch.distributed as dist
    from sdv_loam_amd import synthetic as syn
    from sdv_loam_amd.parallel import ShardedEnergyFunctional
    W = syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=3, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    W.idepth_zero = (W.idepth + np.random.default_rng(1).normal(0, 2e-4, W.nP)).astype(np.float32)
    if not direct:
        monkeypatch.setenv("SDVGN_NO_DIRECT_RCCL", "1")
    if not dist.is_initialized():
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29743 + int(direct)), rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        S = ShardedEnergyFunctional(W, 0, 1, 0, force_collective=True)
        assert S.direct_rccl == direct
        ts = S.optimize(6, want_trace=True, fixed_its=True)
        G = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        tg = G.optimize(6, fixed_its=True)
        assert tg[0, 2] == 0 and len(tg) == 6                                 # first step rejected; five more bodies the speculative way
        assert np.array_equal(ts, tg)
        for a, b in zip(S.ef.state(), G.state()):
            assert np.array_equal(a, b)
        assert np.array_equal(S.ef.points(), G.points())
        with torch.cuda.stream(S.stream):
            xs = S.ef.solveSystemF(3, 0.1)
        assert np.array_equal(xs, G.solveSystemF(3, 0.1))
        del S
    finally:
        dist.destroy_process_group()


def test_fence_instruments_catch_stray_accesses():
    """The allocation instruments the fault hunts rest on (csrc/devmem.hpp) are only evidence if a stray access really ends the process: one
    element past the end / before the start of a fenced buffer, 3 MB past the end under 64 MB guards, and a read through a stale pointer
    after the free under SDVGN_GUARD_QUARANTINE=1 must all fail loudly; accesses inside a buffer must not (tools/probe_fence.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "probe_fence.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("nF", [5, 6, 7])
def test_full_size_linearize_other_window_sizes(api, orc, nF):
    """k_ef_linearize's XCD-aware work order (target-major eighths of the grid) on windows of 5, 6 and 7 key-frames x 2000 points at KITTI
    resolution: Jacobians, states and energies bit-identical to the oracle, and one solve on top."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=nF, pts_per_kf=2000, seed=10 + nF, calib=syn.KITTI00)
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    check_solve(G, O, 0, 0.1)


@pytest.mark.parametrize("nF", [2, 3])
def test_small_windows(api, orc, nF):
    """Windows of 2 and 3 key-frames (system dimension 16 and 22: the blocked LDL^T ends on a 4- and a 2-column panel; optimize runs its
    100 / 75 iteration budgets, FullSystemOptimize.cpp:349-350): linearise, solve and the whole loop against the oracle."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=320, h=160, nF=nF, pts_per_kf=150, seed=30 + nF, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5), state_sigma=1e-3, idepth_sigma=0.01)
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    check_solve(G, O, 0, 0.1)
    check_solve(G, O, 2, 1e-3)
    G2, O2 = pair(api, orc, W)
    tg, to = G2.optimize(), O2.optimize()
    assert len(tg) == len(to) and np.array_equal(tg[:, 2], to[:, 2])
    assert rel_err(G2.state()[1], O2.state()[1]) < 1e-4


def test_many_points_per_host(api, orc):
    """More than 4096 points on one host key-frame: the accumulate falls back to its two-launch form (several 64-point tiles per Schur-Gram
    workgroup) and the reduce sums 64 chunks per host (16 per quarter) -- same results as the oracle."""
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=640, h=240, nF=3, pts_per_kf=4500, seed=41, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    check_solve(G, O, 0, 0.1)


@pytest.mark.parametrize("scale", [1e9, 3e10])
def test_indefinite_system_pivoted_fallback(api, orc, window, scale):
    """Eigen's pivoted `HFinalScaled.ldlt().solve()` (EnergyFunctional.cpp:743) returns a finite x on an indefinite system.  An
    indefinite marginalisation prior (2x2 block [[1, 2], [2, 1]] * scale between two pose coordinates: eigenvalues 3, -1) makes the stitched
    system indefinite with a positive diagonal; the device's unpivoted factorisation meets a negative pivot and must fall back to the pivoted
    one (status 2) with the oracle's x -- and the optimize loop on that window must follow the oracle's decisions."""
    W = copy.copy(window)
    n = 4 + 6 * W.nF
    HM = np.array(W.HM, np.float64).copy()
    i, j = 4 + 6 * 1 + 1, 4 + 6 * 3 + 2
    HM[i, :] = HM[:, i] = 0; HM[j, :] = HM[:, j] = 0
    HM[i, i] = HM[j, j] = scale
    HM[i, j] = HM[j, i] = 2 * scale
    W.HM = HM
    G, O = pair(api, orc, W)
    check_linearize(G, O)
    G.applyRes(); O.applyRes()
    xg = G.solveSystemF(0, 0.1)
    O.solveSystemF(0, 0.1)
    so = O.system()
    assert np.linalg.eigvalsh(so["HFinal"]).min() < 0 and np.all(np.diag(so["HFinal"]) > 0)     # indefinite, positive diagonal
    assert G.solve_status() == 2
    assert np.all(np.isfinite(xg)) and rel_err(xg, so["x"]) < 1e-4        # (the accumulators themselves agree to 1e-5)
    assert rel_err(G.points()[:, 8], O.points()[:, 8]) < 1e-4
    # a definite window right after it goes back to the fast path
    G2, O2 = pair(api, orc, window)
    G2.linearizeAll(); G2.applyRes(); G2.solveSystemF(0, 0.1)
    assert G2.solve_status() == 0
    # whole loop on the indefinite window
    G, O = pair(api, orc, W)
    check_optimize(G, O, 5)


def test_arith_mode_tolerance(api, orc):
    """sdvgn_ef_set_arith(1): linearize with fused multiply-adds and 1-ulp reciprocals / square roots.  Against the exact mode (and through
    it the oracle): J and energies to float rounding, residual states identical except at threshold ties, the loop's decisions identical
    and the increments within BASELINE.json's 1e-4 -- at the named window size."""
    from sdv_loam_amd import synthetic as syn
    W = low_thresholds(syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=syn.KITTI00, state_sigma=1e-3, idepth_sigma=0.01))
    Ge = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    Gf = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    Gf.set_arith(1)
    ee, ef = Ge.linearizeAll(), Gf.linearizeAll()
    se, sf = Ge.residual_state(), Gf.residual_state()
    Je, Jf = Ge.residual_J(0), Gf.residual_J(0)
    differ = se["new_state"] != sf["new_state"]
    assert differ.mean() < 1e-4                                                   # ties at the IN / OUTLIER threshold only
    same = ~differ & (se["new_state"] != 1)
    assert not np.array_equal(Je[same], Jf[same])                                 # (it IS different arithmetic)
    assert np.abs(Je[same] - Jf[same]).max() <= 2e-5 * np.abs(Je[same]).max()
    # photometric energies: a last-place change of an interpolated intensity (~150) against residuals of a few grey levels
    # (the few beyond that sit on a tie of the per-pixel tests -- a pattern pixel on the image-border bound or at the Huber knee)
    off = ~np.isclose(se["energy_with_outlier"][same], sf["energy_with_outlier"][same], rtol=3e-4, atol=1e-3)
    assert off.mean() < 2e-3, off.mean()
    assert rel_err(ef, ee) < 1e-6
    O = __import__("oracle.backend", fromlist=["OracleEF"]).OracleEF(W.w, W.h).load(W)
    Ge.load(W); Gf.load(W)
    te, tf, to = Ge.optimize(6), Gf.optimize(6), O.optimize(6)
    assert len(tf) == len(to) and np.array_equal(tf[:, 2], to[:, 2])              # same accept / reject decisions as the oracle
    n = Ge.dim
    errs = [rel_err(tf[i, 7:7 + n], to[i, 7:7 + n]) for i in range(len(to))]
    # BASELINE's contract: the increment of a Gauss-Newton iteration within 1e-4 of the CPU path's -- from the SAME linearisation point, i.e. the
    # call's first body.  Later bodies start from states that already differ in the last places, and one residual that crosses the IN / OUTLIER
    # threshold the other way (a tie) moves an increment by a few 1e-4 of its norm: bounded an order of magnitude wider
    assert errs[0] < 1e-4, errs
    assert max(errs) < 2e-3, errs
    assert rel_err(Gf.state()[1], O.state()[1]) < 1e-3 and rel_err(Gf.state()[2], O.state()[2]) < 1e-4, (errs, rel_err(Gf.state()[1], O.state()[1]), rel_err(Gf.state()[2], O.state()[2]))


@pytest.mark.parametrize("threads", [True, False])
def test_optimize_batch_side_by_side(api, orc, threads, monkeypatch):
    """sdvgn_ef_optimize_batch, both forms -- one launch sequence for all windows (the default, backend_lockstep.inc) and B windows on their own
    streams and host threads (SDVGN_BATCH_THREADS=1) -- every window ends exactly where its own sdvgn_ef_optimize call ends (bit for bit),
    whatever runs beside it."""
    from sdv_loam_amd import synthetic as syn
    if threads:
        monkeypatch.setenv("SDVGN_BATCH_THREADS", "1")
    else:
        monkeypatch.delenv("SDVGN_BATCH_THREADS", raising=False)
    Ws = [low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=20 + k, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5),
                                         state_sigma=1e-3, idepth_sigma=0.01)) for k in range(6)]
    solo = []
    for W in Ws:
        G = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
        tr = G.optimize(6)
        solo.append((len(tr), G.state(), G.residual_state()))
    for rep in range(3):
        Gs = [api.EnergyFunctional(W.w, W.h, max_points=W.nP, stream=api.EnergyFunctional.STREAM_OWN).load(W) for W in Ws]
        its = api.optimize_batch(Gs, 6)
        for G, n, (n0, st0, rs0) in zip(Gs, its, solo):
            assert n == n0
            st = G.state()
            assert np.array_equal(st[0], st0[0]) and np.array_equal(st[1], st0[1]) and np.array_equal(st[2], st0[2])
            rs = G.residual_state()
            assert np.array_equal(rs["state"], rs0["state"]) and np.array_equal(rs["active"], rs0["active"])
    # a handle twice in one batch is refused
    import ctypes as C
    arr = (C.c_void_p * 2)(Gs[0].h_, Gs[0].h_)
    assert Gs[0].L.sdvgn_ef_optimize_batch(C.cast(arr, C.c_void_p), 2, 6, 0, None) < 0


def _lock_windows(syn, kind):
    cal = dict(fx=400., fy=410., cx=319.5, cy=119.5)
    if kind == "mixed":      # different seeds AND different shapes in one call: 5 / 4 / 3 / 6 key-frames, 300 / 200 / 150 / 700 points per frame
        specs = [dict(nF=5, pts_per_kf=300, seed=2), dict(nF=4, pts_per_kf=200, seed=3, state_sigma=1e-3, idepth_sigma=0.01),
                 dict(nF=3, pts_per_kf=150, seed=4, state_sigma=1e-3, idepth_sigma=0.01), dict(nF=6, pts_per_kf=700, seed=5, state_sigma=1e-3, idepth_sigma=0.01),
                 dict(nF=5, pts_per_kf=300, seed=3, state_sigma=1e-3, idepth_sigma=0.01)]
    else:                    # the same shape, different data
        specs = [dict(nF=5, pts_per_kf=300, seed=2)] + [dict(nF=5, pts_per_kf=300, seed=20 + k, state_sigma=1e-3, idepth_sigma=0.01) for k in range(7)]
    return [low_thresholds(syn.make_window(w=640, h=240, calib=cal, **s)) for s in specs]


def _same_window_result(A, B, solved=True):
    for x, y in zip(A.state(), B.state()):
        assert np.array_equal(x, y)
    sa, sb = A.residual_state(), B.residual_state()
    for k in sa:
        assert np.array_equal(sa[k], sb[k]), k
    assert np.array_equal(A.frame_energy_th(), B.frame_energy_th())
    if solved:                                   # (the per-point planes are the outputs of an accumulate: none has run in a call of zero bodies)
        assert np.array_equal(A.points(), B.points())
    assert A.accepted_steps() == B.accepted_steps()


@pytest.mark.parametrize("kind,its,fixed", [("mixed", 12, True), ("mixed", 6, False), ("same", 6, True), ("same", 1, True), ("same", 0, True)])
def test_lockstep_equals_own_optimize_call(api, orc, kind, its, fixed):
    """sdvgn_ef_optimize_lockstep: B windows as ONE launch sequence, accept / reject, lambda and the choice of state copies per window in device
    memory.  Every window must end where its own sdvgn_ef_optimize call ends, bit for bit -- trace rows (lambda, verdict, the three energies,
    canbreak, x, the newest frame's threshold), final states, per-residual planes, point planes, thresholds, and what the next solve returns --
    with rejected steps, consecutive rejections, a rejection in the last body, windows of different shapes, and loops that end early."""
    from sdv_loam_amd import synthetic as syn
    Ws = _lock_windows(syn, kind)
    solo = [api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for W in Ws]
    tr_solo = [G.optimize(its, fixed_its=fixed) for G in solo]
    if kind == "mixed" and fixed:
        rej = tr_solo[0][:, 2] == 0
        assert rej.any() and (~rej).any() and (rej[:-1] & rej[1:]).any()
    Gs = [api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for W in Ws]
    n, tr = api.optimize_lockstep(Gs, its, fixed_its=fixed)
    for b, (G, S) in enumerate(zip(Gs, solo)):
        assert n[b] == len(tr_solo[b]), (b, n, [len(t) for t in tr_solo])
        assert np.array_equal(tr[b], tr_solo[b]), (b, np.argwhere(tr[b] != tr_solo[b])[:8])
        _same_window_result(G, S, solved=its > 0)
    # the handles carry on like after a call of their own: the next solve, a second call (lock-step against single), the loop's tail
    for G, S in zip(Gs, solo):
        assert np.array_equal(G.solveSystemF(3, 0.1), S.solveSystemF(3, 0.1))
    n2, tr2 = api.optimize_lockstep(Gs, 4, fixed_its=True)
    for b, (G, S) in enumerate(zip(Gs, solo)):
        assert np.array_equal(tr2[b], S.optimize(4, fixed_its=True)), b
        _same_window_result(G, S)
        ea, eb = G.optimize_finish(), S.optimize_finish()
        assert ea[0] == eb[0] and all(np.array_equal(x, y) for x, y in zip(ea[1:], eb[1:]))


def test_lockstep_full_size_and_batch_entry(api, orc):
    """At the named window size (8 x 2000 points, 112 000 residuals): four windows through sdvgn_ef_optimize_batch -- which takes the lock-step
    path for them -- against their own calls; and with linearised residuals in the window (the point part of calcLEnergy is not 0)."""
    from sdv_loam_amd import synthetic as syn
    Ws = [low_thresholds(syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=k, calib=syn.KITTI00, state_sigma=1e-3, idepth_sigma=0.01)) for k in range(4)]
    solo = [api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for W in Ws]
    n_solo = [len(G.optimize(6, fixed_its=True)) for G in solo]
    Gs = [api.EnergyFunctional(W.w, W.h, max_points=W.nP, stream=api.EnergyFunctional.STREAM_OWN).load(W) for W in Ws]
    assert api.optimize_batch(Gs, 6, fixed_its=True) == n_solo
    for G, S in zip(Gs, solo):
        _same_window_result(G, S)
    # the key-frame cycle in between: fix the linearisation of a third of the points, then optimise again
    for G, S, W in zip(Gs, solo, Ws):
        mask = (np.arange(W.nP) % 3 == 0).astype(np.uint8)
        G.fixLinearization(mask); S.fixLinearization(mask)
    n, tr = api.optimize_lockstep(Gs, 5, fixed_its=True)
    for b, (G, S) in enumerate(zip(Gs, solo)):
        assert np.array_equal(tr[b], S.optimize(5, fixed_its=True)), b
        _same_window_result(G, S)


def test_lockstep_grid_larger_than_the_chip(api, orc):
    """k_lock_tail at B = 16 windows of the named size is a grid of 2 112 workgroups of 1 024 lanes -- four times what is resident at once -- in
    which the resubstitute / step workgroups of every window spin on words its factorisation workgroup publishes: that ends only because the
    factorisations (workgroup ids 0 .. B-1) are dispatched first (csrc/backend_lockstep.inc, comment at k_lock_tail).  All 16 loops must run to
    the end without a give-up (no SDVGN_E_STATE) and leave every window where its own sdvgn_ef_optimize call leaves it, bit for bit."""
    from sdv_loam_amd import synthetic as syn
    Ws = [low_thresholds(syn.make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=k, calib=syn.KITTI00, state_sigma=3e-3, idepth_sigma=0.02)) for k in range(2)]
    solo = [api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for W in Ws]
    tr_solo = [G.optimize(6, fixed_its=True) for G in solo]
    Gs = [api.EnergyFunctional(Ws[b % 2].w, Ws[b % 2].h, max_points=Ws[b % 2].nP).load(Ws[b % 2]) for b in range(16)]
    n, tr = api.optimize_lockstep(Gs, 6, fixed_its=True)
    for b, G in enumerate(Gs):
        assert n[b] == 6 and np.array_equal(tr[b], tr_solo[b % 2]), b
        _same_window_result(G, solo[b % 2])


def test_two_lockstep_calls_at_a_time(api, orc):
    """Two host threads, each driving a lock-step call on its own set of windows of ONE device at the same time (two launch-sequence pools per
    device, csrc/backend_lockstep.inc `lock_pool_acquire`): every window ends where its own sdvgn_ef_optimize call leaves it, bit for bit."""
    import threading
    from sdv_loam_amd import synthetic as syn
    Ws = [low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=20 + k, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5), state_sigma=2e-3,
                                         idepth_sigma=0.02)) for k in range(2)]
    solo = [api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W) for W in Ws]
    tr_solo = [G.optimize(6, fixed_its=True) for G in solo]
    for rep in range(3):
        Gs = [api.EnergyFunctional(Ws[b % 2].w, Ws[b % 2].h, max_points=Ws[b % 2].nP).load(Ws[b % 2]) for b in range(6)]
        out = {}

        def run(key, hs):
            out[key] = api.optimize_lockstep(hs, 6, fixed_its=True)
        th = [threading.Thread(target=run, args=(0, Gs[:3])), threading.Thread(target=run, args=(1, Gs[3:]))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        for key, hs in ((0, Gs[:3]), (1, Gs[3:])):
            n, tr = out[key]
            for j, G in enumerate(hs):
                b = 3 * key + j
                assert n[j] == 6 and np.array_equal(tr[j], tr_solo[b % 2]), (rep, b)
                _same_window_result(G, solo[b % 2])


def test_lockstep_refuses_what_it_does_not_take(api, orc, window):
    import ctypes as C
    G = api.EnergyFunctional(window.w, window.h, max_points=window.nP).load(window)
    arr = (C.c_void_p * 2)(G.h_, G.h_)
    assert G.L.sdvgn_ef_optimize_lockstep(C.cast(arr, C.c_void_p), 2, 6, 0, None, None, 0, 0) < 0        # a handle twice
    arr1 = (C.c_void_p * 1)(G.h_)
    assert G.L.sdvgn_ef_optimize_lockstep(C.cast(arr1, C.c_void_p), 1, 6, 2, None, None, 0, 0) < 0       # the literal re-linearisation: sdvgn_ef_optimize only
    assert G.L.sdvgn_ef_optimize_lockstep(C.cast(arr1, C.c_void_p), 1, 6, 0, None, None, 0, 0) == 0


@pytest.mark.parametrize("seed,its", [(2, 12), (3, 6), (4, 7)])
def test_fused_apply_single_window_is_bit_identical(api, orc, seed, its, monkeypatch):
    """The loop's default since round 6 (B): the linearise leaves applyRes in the second copies of the flag / state / energy / JpJd planes, an accepted step swaps
    pointers, and the accept test is a workgroup of the next body's accumulate launch (k_ef_acc_stats) -- against SDVGN_FUSED_APPLY=0 (A: apply workgroups in a
    statistics launch of its own, the form of rounds 3-5) and against the fused form with the statistics as their own launch (C: SDVGN_DEBUG_FLAGS bit 9):
    same trace, states, planes, next solve and loop tail, bit for bit; also after operations that write the planes in place between two calls (fixLinearization)."""
    from sdv_loam_amd import synthetic as syn
    kw = {} if seed == 2 else dict(state_sigma=1e-3, idepth_sigma=0.01)
    W = low_thresholds(syn.make_window(w=640, h=240, nF=5, pts_per_kf=300, seed=seed, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5), **kw))
    monkeypatch.delenv("SDVGN_FUSED_APPLY", raising=False)
    monkeypatch.delenv("SDVGN_DEBUG_FLAGS", raising=False)
    A = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    B = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    monkeypatch.setenv("SDVGN_DEBUG_FLAGS", "512")
    Cw = api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)      # (the flags are read when the window is loaded)
    monkeypatch.delenv("SDVGN_DEBUG_FLAGS")
    monkeypatch.setenv("SDVGN_FUSED_APPLY", "0")
    ta = A.optimize(its, fixed_its=True)
    monkeypatch.delenv("SDVGN_FUSED_APPLY")
    tb = B.optimize(its, fixed_its=True)
    tc = Cw.optimize(its, fixed_its=True)
    assert np.array_equal(ta, tb) and np.array_equal(ta, tc)
    assert 0 < ta[:, 2].sum() < len(ta) or seed != 2                   # (seed 2: accepted and rejected steps both occur)
    # the default loop really took the fast forms (a silent fall-back would only show as a slower bench): every body but the last queues its successor's accumulate
    # ahead of the verdict with the accept test inside it; the two older forms never do the latter
    la_b, _, merged_b, pre_b = B.loop_counters()
    assert la_b == its - 1 and merged_b == its - 1 and pre_b == its - 1, B.loop_counters()
    assert A.loop_counters()[2] == 0 and Cw.loop_counters()[2] == 0 and A.loop_counters()[3] == its - 1
    _same_window_result(A, B)
    _same_window_result(A, Cw)
    assert np.array_equal(A.residual_J(0), B.residual_J(0))
    mask = (np.arange(W.nP) % 3 == 0).astype(np.uint8)
    A.fixLinearization(mask); B.fixLinearization(mask)
    tb2 = B.optimize(5, fixed_its=True)
    monkeypatch.setenv("SDVGN_FUSED_APPLY", "0")
    ta2 = A.optimize(5, fixed_its=True)
    monkeypatch.delenv("SDVGN_FUSED_APPLY")
    assert np.array_equal(ta2, tb2)
    _same_window_result(A, B)
    ea, eb = A.optimize_finish(), B.optimize_finish()
    assert ea[0] == eb[0] and all(np.array_equal(x, y) for x, y in zip(ea[1:], eb[1:]))
