"""The HIP path (through the C ABI) against the REFERENCE'S OWN code.

oracle/_ref/libref.so -- the reference's FullSystem / OptimizationBackend translation units compiled unmodified (oracle/Makefile target
`ref`; built where /root/reference exists, travels to the GPU box as a prebuilt file) -- is driven by RefEF / RefTracker with the same
inputs as the product.  The checks are the ones tests/test_backend_gpu.py and tests/test_tracker_gpu.py run against the CPU oracle
(imported from there), so every tolerance is the same: per-residual / per-point float terms, states and counters bit-exact; reduced
quantities 1e-5; increments 1e-4 (BASELINE.json north_star)."""
import copy

import numpy as np
import pytest

from common import load_problem, rel_err, small_problem, start_pose

pytestmark = pytest.mark.gpu


def _have_ref():
    from oracle import refpin
    L = refpin.ref_lib()
    return L is not None and hasattr(L, "ref_ef_create") and hasattr(L, "ref_tracker_create")


needs_ref = pytest.mark.skipif(not _have_ref(), reason="oracle/_ref/libref.so not present on this machine")


@pytest.fixture(scope="module")
def bapi(sdvgn_lib):
    from sdv_loam_amd import backend_api
    return backend_api


@pytest.fixture(scope="module")
def tapi(sdvgn_lib):
    from sdv_loam_amd import api as A
    return A


def _pair_ef(bapi, W):
    from oracle.backend import RefEF
    G = bapi.EnergyFunctional(W.w, W.h, max_points=max(W.nP, 16)).load(W)
    R = RefEF(W.w, W.h).load(W)
    return G, R


@needs_ref
@pytest.mark.parametrize("cfg", [dict(w=640, h=240, nF=5, pts_per_kf=300, seed=2, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)),
                                 dict(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0)])
def test_backend_linearize_solve_vs_reference(bapi, orc, cfg):
    """b1-b7 at a small window and at BASELINE.json configs[2] (8 x 2000 points, 112 000 residuals)"""
    from sdv_loam_amd import synthetic as syn
    from test_backend_gpu import check_linearize, check_solve
    W = syn.make_window(**cfg)
    G, R = _pair_ef(bapi, W)
    G.compute_nullspaces(); R.compute_nullspaces()
    check_linearize(G, R)                      # J, energies, states bit-identical with PointFrameResidual::linearize; exact threshold
    G.applyRes(); R.applyRes()
    sg, sr = G.residual_state(), R.residual_state()
    assert np.array_equal(sg["state"], sr["state"]) and np.array_equal(sg["active"], sr["active"])
    check_solve(G, R, 0, 0.1)                  # accumulators, stitched system, x, point steps
    check_solve(G, R, 3, 1e-3)


@needs_ref
@pytest.mark.parametrize("cfg", [dict(w=640, h=240, nF=5, pts_per_kf=300, seed=3, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5)),
                                 dict(w=1241, h=376, nF=7, pts_per_kf=2000, seed=7, state_sigma=1e-3, idepth_sigma=0.01)])
def test_optimize_vs_reference(bapi, orc, cfg):
    """sdvgn_ef_optimize + sdvgn_ef_optimize_finish against the reference's FullSystem::optimize (the whole function)"""
    from sdv_loam_amd import synthetic as syn
    from test_backend_gpu import low_thresholds
    W = low_thresholds(syn.make_window(**cfg))
    G, R = _pair_ef(bapi, W)
    G.compute_nullspaces(); R.compute_nullspaces()
    tg = G.optimize(6)
    e, rb, ng, rm = G.optimize_finish()
    rmse, steps, removed, log = R.optimize_full(6)
    assert len(tg) == len(steps) and [bool(a) for a in tg[:, 2]] == [s[0] for s in steps]           # accept / reject sequence
    assert np.allclose(tg[:, 3], [s[2] for s in steps], rtol=1e-5, atol=2e-3)                       # energies as the reference prints them
    vg, sg, ig = G.state()
    vr, sr, ir = R.state()
    assert np.allclose(vg, vr, rtol=1e-9) and rel_err(sg, sr) < 1e-4 and rel_err(ig, ir) < 1e-6
    assert np.array_equal(rm, removed)
    assert np.allclose(G.frame_energy_th(), R.frame_energy_th(), rtol=1e-4)
    assert abs(np.sqrt(e / R.resInA()) - rmse) <= 1e-5 * rmse
    prb, png = R.point_stats()
    assert np.array_equal(ng, png) and np.allclose(rb, prb, rtol=1e-4, atol=1e-6)
    keep = removed == 0
    rg, rr = G.residual_state(), R.residual_state()
    assert np.array_equal(rg["state"][keep], rr["state"][keep]) and np.array_equal(rg["active"][keep], rr["active"][keep])


@needs_ref
@pytest.mark.parametrize("seed,n", [(0, 400), (2, 2000)])
def test_tracker_res_gs_vs_reference(tapi, orc, seed, n):
    """a1, a2, a4-a6 against CoarseTracker::makeK / calcRes / calcGSSSE and FrameHessian::makeImages"""
    from test_tracker_gpu import check_res_gs
    P = small_problem(seed=seed, n=n, noise=2.0)
    G = load_problem(tapi.CoarseTracker(P.w, P.h, P.levels, max_points=1 << 16, max_batch=40), P, ref_aff=(0.01, 1.0))
    R = load_problem(orc.RefTracker(P.w, P.h, P.levels), P, ref_aff=(0.01, 1.0))
    pose = start_pose(orc, P, seed)
    for lvl in range(P.levels):
        kg, kig = G.get_K(lvl)
        kr, kir = R.get_K(lvl)
        assert np.array_equal(kg, kr) and np.array_equal(kig, kir)
        g, r = G.get_pyr(lvl), R.get_pyr(lvl)
        assert np.array_equal(g[..., 0], r[..., 0]) and np.array_equal(g[1:-1, :, 1:], r[1:-1, :, 1:])
        check_res_gs(G, R, lvl, pose, 0.03, 2.0, 20.0)
        check_res_gs(G, R, lvl, P.gt_pose, 0.04, 2.5, 20.0)


@needs_ref
@pytest.mark.parametrize("full", [False, True])
def test_track_vs_reference(tapi, orc, full):
    """a7 against CoarseTracker::trackNewestCoarse: host-driven and device-resident LM; the second case is BASELINE.json configs[1]"""
    from sdv_loam_amd import synthetic as syn
    if full:
        P = syn.make_tracker_problem(w=1241, h=376, levels=4, n_points=2000, seed=0, calib=syn.KITTI00,
                                     gt_xi=[0.03, -0.02, 0.05, 0.004, -0.006, 0.002], gt_aff=(0.03, 1.5))
    else:
        P = small_problem(seed=1, n=600, noise=1.0)
    G = load_problem(tapi.CoarseTracker(P.w, P.h, P.levels, max_points=1 << 16, max_batch=4), P)
    R = load_problem(orc.RefTracker(P.w, P.h, P.levels), P)
    for seed in (0, 1, 2):
        start = start_pose(orc, P, seed, 0.02, 0.003)
        okr, pr, ar, lrr, flr, _ = R.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
        dr = orc.se3_log(orc.se3_mul(pr, orc.se3_inverse(start)))
        okg, pg, ag, lrg, flg, _ = G.trackNewestCoarse(start, (0.0, 0.0), P.levels - 1)
        dg = orc.se3_log(orc.se3_mul(pg, orc.se3_inverse(start)))
        assert okg == okr and rel_err(dg, dr) < 1e-4 and np.allclose(ag, ar, rtol=1e-4, atol=1e-6)
        assert np.allclose(lrg, lrr, rtol=1e-4, atol=1e-4, equal_nan=True)     # (noise-free problem: the final RMSE is rounding noise, ~2e-4)
        okb, pb, ab, _, _ = G.trackBatch(np.stack([start, start]), np.zeros((2, 2)), P.levels - 1)
        db = orc.se3_log(orc.se3_mul(pb[1], orc.se3_inverse(start)))
        assert bool(okb[1]) == okr and rel_err(db, dr) < 1e-4


# ---- rows f1, f2, f4 and the marginalisation step: the HIP path against the reference's own members directly (round 4; until then these
# rows were HIP vs oracle with the oracle pinned to the reference on the CPU) -----------------------------------------------------------------

@needs_ref
@pytest.mark.parametrize("seed,kw", [(0, {}), (1, dict(pose_err=(0.3, 0.02))), (3, dict(pose_err=(0.01, 0.001), outlier_frac=0.2))])
def test_struct_pose_vs_reference(tapi, orc, seed, kw):
    """f1 against CoarseTracker::structPoseEstimation / calcHandb / calculateRes (CoarseTracker.cpp:840-1007)"""
    from sdv_loam_amd import synthetic as syn
    S = syn.make_struct_problem(n=1200, seed=seed, **kw)
    G = tapi.CoarseTracker(S.w, S.h, 4, max_points=2048)
    R = orc.RefTracker(S.w, S.h, 4)
    for T in (G, R):
        T.makeK(**S.calib)
    args = (S.u, S.v, S.idepth, S.host_idx, S.host_poses7, S.obs)
    w2c = orc.se3_inverse(S.init_curToWorld7)
    Hg, bg, eg, ng = G.structResHb(w2c, *args)
    Hr, br, er, nr = R.structResHb(w2c, *args)
    assert ng == nr and rel_err(Hg, Hr) < 1e-5 and rel_err(bg, br) < 1e-5 and abs(eg - er) <= 1e-5 * er
    pg, _, _ = G.structPoseEstimation(S.init_curToWorld7, *args)
    pr, _, _ = R.structPoseEstimation(S.init_curToWorld7, *args)
    dg = orc.se3_log(orc.se3_mul(orc.se3_inverse(S.init_curToWorld7), pg))
    dr = orc.se3_log(orc.se3_mul(orc.se3_inverse(S.init_curToWorld7), pr))
    assert rel_err(dg, dr) < 1e-4


@needs_ref
@pytest.mark.parametrize("seed,pose_err,edge", [(2, (0.01, 0.001), 0.3), (5, (0.05, 0.006), 0.5)])
def test_reprojector_vs_reference(orc, sdvgn_lib, seed, pose_err, edge):
    """f2 against Reprojector::reprojectPoint / findMatchDirect / align1D / align2D (Reprojector.cpp)"""
    from oracle.reproject import RefReprojector
    from sdv_loam_amd import reproject_api, synthetic as syn
    import test_reproject_gpu as T
    P = syn.make_reproject_problem(T._window(seed), levels=3, seed=seed, pose_err=pose_err, edgelet_frac=edge)
    R = RefReprojector(P.w, P.h, P.levels)
    G = reproject_api.Reprojector(P.w, P.h, P.levels, max_frames=8, max_points=4096)
    for X in (R, G):
        X.set_calib(**P.calib)
        for k in range(len(P.frame_poses7)):
            X.set_frame(k, P.frame_poses7[k], P.frame_images[k], 1.0, 0.0, 0.0)
        X.set_cur(P.cur_pose7, P.cur_pyr, 1.0, 0.0, 0.0)
    g, good = T._compare(P, G, R)                      # cells, qualities, success flags, levels exact; matched positions bit-identical
    assert good.sum() > 0.3 * P.n


@needs_ref
@pytest.mark.parametrize("seed,pose_err", [(2, (0.0, 0.0)), (4, (0.1, 0.01))])
def test_trace_on_vs_reference(tapi, orc, seed, pose_err):
    """f4 (part 1) against ImmaturePoint::traceOn (ImmaturePoint.cpp:47-353): all six outputs bit-identical"""
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    import test_trace_gpu as T
    W = T._window(seed)
    P = syn.make_trace_problem(W, target=2, pose_err=pose_err, seed=seed)
    G = T._gpu(P)
    sr = trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status, reference=True)
    sg = G.tracePoints(P.KRKi, P.Kt, P.aff, P.idepth_min, P.idepth_max, P.quality, P.status)
    T._same(sg, sr)
    assert (sr["status"] == syn.IPS_GOOD).sum() > 0.3 * P.n


@needs_ref
@pytest.mark.parametrize("seed,rel", [(2, 0.2), (4, 0.6)])
def test_optimize_immature_vs_reference(bapi, orc, seed, rel):
    """f4 (part 2) against FullSystem::optimizeImmaturePoint + ImmaturePoint::linearizeResidual (FullSystemOptPoint.cpp:18-185)"""
    from oracle.backend import RefEF
    from sdv_loam_amd import synthetic as syn
    import test_immature_gpu as T
    W = syn.make_window(w=320, h=200, nF=4, pts_per_kf=250, seed=seed, calib=T.CAL)
    G = bapi.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    R = RefEF(W.w, W.h).load(W)
    imin, imax, eth = T._args(W, seed, rel)
    a = (W.host, W.u, W.v, imin, imax, eth, W.color, W.weights, W.isFromSensor)
    rg, rr = G.optimizeImmature(*a), R.optimizeImmature(*a)
    T._same(rg, rr)                                    # result codes, inverse depths, residual states bit-identical
    assert (rr[0] == 1).sum() > 0.3 * W.nP


@needs_ref
def test_marginalization_vs_reference(bapi, orc):
    """b2 mode 2 against EFResidual::fixLinearizationF, EnergyFunctional::marginalizePointsF / dropPointsF / marginalizeFrame
    (EnergyFunctionalStructs.cpp:45-55, EnergyFunctional.cpp:434-597)"""
    from oracle.backend import RefEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=640, h=240, nF=8, pts_per_kf=300, seed=9, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))
    W.idepth_zero = (W.idepth + np.random.default_rng(5).normal(0, 2e-4, W.nP)).astype(np.float32)   # deltaF != 0
    G, R = _pair_ef(bapi, W)
    for X in (G, R):
        X.linearizeAll(); X.applyRes()
    rng = np.random.default_rng(1)
    marg = (rng.random(W.nP) < 0.2).astype(np.uint8)
    drop = ((rng.random(W.nP) < 0.1) & (marg == 0)).astype(np.uint8)
    G.fixLinearization(marg); R.fixLinearization(marg)
    rg, lg = G.res_toZero(); rr, lr = R.res_toZero()
    # (the reference forms the three inner products of fixLinearizationF with Eigen's .dot(), whose reduction order the device's sequential
    # sums do not repeat: last-digit differences; the oracle reproduces Eigen's order when asked to -- tests/test_ref_pin_backend.py)
    assert np.array_equal(lg, lr) and lr.sum() > 20 and np.allclose(rg, rr, rtol=2e-5, atol=2e-6)
    H0, b0 = R.marg_prior()
    for idx in (0, 3):                                  # marginalizeFrame on the prior as it is now (a pure function of HM, bM, the frame prior)
        Hg, bg = G.marginalizeFrame(idx)
        Hr, br = R.marginalizeFrame(idx)
        assert rel_err(Hg, Hr) < 1e-6 and rel_err(bg, br) < 1e-6
    G.marginalizePoints(marg, drop); R.marginalizePoints(marg, drop)
    Hg, bg = G.marg_prior(); Hr, br = R.marg_prior()
    assert np.linalg.norm(Hr - H0) > 0
    assert rel_err(Hg - H0, Hr - H0) < 1e-5 and rel_err(bg - b0, br - b0) < 1e-5 and rel_err(Hg, Hr) < 1e-6
    for idx in (1, W.nF - 1):
        Hgf, bgf = G.marginalizeFrame(idx)
        Hrf, brf = R.marginalizeFrame(idx)
        assert rel_err(Hgf, Hrf) < 1e-6 and rel_err(bgf, brf) < 1e-6
    xg = G.solveSystemF(0, 0.1); R.solveSystemF(0, 0.1)
    assert rel_err(xg, R.system()["x"]) < 1e-4
