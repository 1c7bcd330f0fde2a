"""The CPU oracle against the REFERENCE'S OWN back-end code.

oracle/_ref/libref.so holds the reference's FullSystem, EnergyFunctional, AccumulatedTopHessianSSE, AccumulatedSCHessianSSE,
PointFrameResidual (Residuals.cpp), FrameFramePrecalc / FrameHessian (HessianBlocks.cpp) ... compiled UNMODIFIED from /root/reference
(oracle/Makefile target `ref`; third-party Eigen / Boost / ROS / OpenCV replaced by oracle/ref_shim, see the header of
oracle/ref_shim/Eigen/Core for the evaluation rules of the Eigen stand-in).  oracle/ref_glue_ef.cpp loads the same flattened window into
the reference's objects; these tests run one window through both and compare:

  SURVEY 8 row   reference function (file:line)                                        agreement asserted here
  b1             PointFrameResidual::linearize / applyRes (Residuals.cpp:60-224,252)    bit-identical J, energies, states
  b2, b3         AccumulatedTopHessianSSE::addPoint<0>, <2> (AccumulatedTopHessian.cpp) bit-identical 13x13 float accumulators, per-point sums
  b7             AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:10-62)     bit-identical accE / accEB / accD / accHcc / accbc
  b4             stitchDoubleMT / stitchDouble                                          <= 1e-12 relative (double, Eigen products)
  b5             setAdjointsF, setDeltaF, FrameFramePrecalc::set (HessianBlocks.cpp:169) float outputs bit-identical except the unused
                                                                                        host == target entries; adjoints <= 1e-14
  b6             solveSystemF + resubstituteF_MT (EnergyFunctional.cpp:650-759,221-282)  x <= 1e-9 relative; point steps bit-identical when
                                                                                        the oracle adds Eigen's 6-term inner product in
                                                                                        Eigen's (halving) order, <= 1e-6 otherwise
  b8             FullSystem::optimize, whole function (FullSystemOptimize.cpp:344-502)  same accept / reject sequence, same energies (to the
                                                                                        printed digits), final states / idepths / thresholds /
                                                                                        removed set identical
  marginalise    fixLinearizationF, marginalizePointsF, marginalizeFrame                res_toZero bit-identical (Eigen order), HM / bM <= 1e-9

Without the library (a machine without /root/reference and without the prebuilt file) the same oracle outputs are checked against the
fixture tests/golden/ref_pin_backend.npz, which holds the reference's outputs for the first window (tools/gen_ref_pin_golden.py).
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_pin_backend.npz")

WINDOWS = [
    dict(w=200, h=96, nF=4, pts_per_kf=60, seed=7, calib=dict(fx=150., fy=152., cx=99.5, cy=47.5)),
    dict(w=320, h=160, nF=7, pts_per_kf=150, seed=3, calib=dict(fx=240., fy=242., cx=159.5, cy=79.5)),
    dict(w=320, h=160, nF=5, pts_per_kf=120, seed=5, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5), state_sigma=1e-3, idepth_sigma=0.01),
]


def _have_ref():
    from oracle import refpin
    L = refpin.ref_lib()
    return L is not None and hasattr(L, "ref_ef_create")


needs_ref = pytest.mark.skipif(not _have_ref(), reason="oracle/_ref/libref.so not built (no /root/reference here)")


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture
def eigen_order(orc):
    """the oracle adds the small Eigen-written dot products in Eigen's own order (oracle/orc_backend.cpp, g_redux_order = 3)"""
    L = orc.lib()
    L.orc_set_redux_order(3)
    yield
    L.orc_set_redux_order(0)


def _pair(cfg):
    from oracle.backend import OracleEF, RefEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(**cfg)
    return W, OracleEF(W.w, W.h).load(W), RefEF(W.w, W.h).load(W)


@needs_ref
@pytest.mark.parametrize("cfg", WINDOWS[:2])
def test_load_setup_precalc_adjoints(orc, cfg):
    W, O, R = _pair(cfg)
    mism, gdiff = R.load_report()
    assert gdiff == 0.0                                  # makeImages' level-0 gradients == the window's (rows 1 .. h-2)
    assert mism == 0                                     # ImmaturePoint's constructor computes the colours / weights the window carries
    for h in range(W.nF):
        for t in range(W.nF):
            a, b = O.precalc(h, t), R.precalc(h, t)
            if h != t:
                assert np.array_equal(a, b), (h, t)      # PRE_KRKiTll, PRE_KtTll, PRE_RTll_0, PRE_tTll_0, PRE_aff_mode, PRE_b0_mode
            else:
                assert np.allclose(a, b, rtol=0, atol=1e-12)   # T * T^-1: cancellation residue around 0 in double (1e-15), never read
    ao, at = O.adjoints()
    ro, rt = R.adjoints()
    assert np.abs(ao - ro).max() <= 1e-14 * np.abs(ro).max() and np.array_equal(at, rt)
    assert np.array_equal(O.adHTdeltaF(), R.adHTdeltaF())
    no, nr = O.compute_nullspaces(), R.compute_nullspaces()
    assert np.abs(no - nr).max() <= 1e-12 * np.abs(nr).max()
    for k in range(W.nF):
        po, pr = O.frame_prior(k), R.frame_prior(k)
        assert np.array_equal(po[0], pr[0]) and np.array_equal(po[1], pr[1])


@needs_ref
@pytest.mark.parametrize("cfg", WINDOWS)
def test_linearize_apply_bit_identical(orc, cfg):
    W, O, R = _pair(cfg)
    eo, er = O.linearizeAll(), R.linearizeAll()          # FullSystem::linearizeAll(false): linearize + setNewFrameEnergyTH
    assert eo == er
    assert np.array_equal(O.residual_J(0), R.residual_J(0))
    so, sr = O.residual_state(), R.residual_state()
    for k in so:
        assert np.array_equal(so[k], sr[k]), k
    assert np.array_equal(O.frame_energy_th(), R.frame_energy_th())
    assert (so["new_state"] == 0).sum() > 0.5 * W.nR and (so["new_state"] == 2).sum() > 0      # inliers and outliers present
    O.applyRes(); R.applyRes()
    assert np.array_equal(O.residual_J(1), R.residual_J(1))                                      # takeDataF
    so, sr = O.residual_state(), R.residual_state()
    for k in so:
        assert np.array_equal(so[k], sr[k]), k


@needs_ref
@pytest.mark.parametrize("cfg", WINDOWS[:2])
def test_solve_system(orc, eigen_order, cfg):
    W, O, R = _pair(cfg)
    for E in (O, R):
        E.compute_nullspaces(); E.resetOOB(); E.linearizeAll(); E.applyRes()
    for it, lam in ((0, 0.1), (3, 1e-3)):                # iteration >= 2: orthogonalize(x) is live (SOLVER_ORTHOGONALIZE_X_LATER)
        O.solveSystemF(it, lam); R.solveSystemF(it, lam)
        assert np.array_equal(O.top_acc(), R.top_acc())                                       # addPoint<0> + AccumulatorApprox
        for a, b in zip(O.sc_acc(), R.sc_acc()):
            assert np.array_equal(a, b)                                                       # SC addPoint + AccumulatorXX / X
        assert O.resInA() == R.resInA()
        so, sr = O.system(), R.system()
        for k in ("HA", "bA", "Hsc", "bsc", "HFinal", "bFinal"):
            assert _rel(so[k], sr[k]) <= 1e-12, k
        assert _rel(so["x"], sr["x"]) <= 1e-9
        po, pr = O.points(), R.points()
        assert np.array_equal(po[:, :8], pr[:, :8])                                           # Hdd / bd / Hcd sums, HdiF, bdSumF
        # the step takes x (double, equal to ~1e-16) through a float cast: identical except where that cast lands on the other side of a
        # rounding boundary
        assert np.abs(po[:, 8] - pr[:, 8]).max() <= 1e-6 * np.abs(pr[:, 8]).max() and (po[:, 8] == pr[:, 8]).mean() > 0.9
        assert _rel(O.frame_steps()[0], R.frame_steps()[0]) <= 1e-9 and _rel(O.frame_steps()[1], R.frame_steps()[1]) <= 1e-9
        assert O.calcLEnergy() == R.calcLEnergy()
        assert abs(O.calcMEnergy() - R.calcMEnergy()) <= 1e-9 * abs(R.calcMEnergy())


@needs_ref
def test_solve_system_default_order_within_tolerance(orc):
    """the oracle's default (left to right) order of the 6-term inner product of resubstituteFPt against the reference: point steps to 1e-6"""
    W, O, R = _pair(WINDOWS[0])
    for E in (O, R):
        E.compute_nullspaces(); E.resetOOB(); E.linearizeAll(); E.applyRes(); E.solveSystemF(0, 0.1)
    po, pr = O.points(), R.points()
    assert np.array_equal(po[:, :8], pr[:, :8])
    assert np.abs(po[:, 8] - pr[:, 8]).max() <= 1e-6 * np.abs(pr[:, 8]).max()


def _optimize_both(cfg, its=6):
    W, O, R = _pair(cfg)
    O.compute_nullspaces(); R.compute_nullspaces()
    tro = O.optimize(its)
    eo, rb, ng, rm = O.optimize_finish()
    rmse, steps, removed, log = R.optimize_full(its)
    return W, O, R, tro, (eo, rb, ng, rm), (rmse, steps, removed, log)


@needs_ref
@pytest.mark.parametrize("cfg", WINDOWS)
def test_full_optimize(orc, eigen_order, cfg):
    """FullSystem::optimize of the reference (loop + tail) against the oracle's optimize + optimize_finish"""
    W, O, R, tro, (eo, rb, ng, rm), (rmse, steps, removed, log) = _optimize_both(cfg)
    assert len(steps) == len(tro) and len(steps) >= 2
    assert [s[0] for s in steps] == [bool(a) for a in tro[:, 2]]                       # accept / reject sequence
    # energies as printed by printOptRes (%f); the reference adds them into an uninitialised Vec10 (FullSystemOptimize.cpp:118), i.e. to
    # whatever the stack held -- ~5e-5 in this build -- so only the printed digits down to 1e-3 are compared
    assert np.allclose([s[2] for s in steps], tro[:, 3], rtol=0, atol=2e-3)
    so, sr = O.state(), R.state()
    assert np.array_equal(so[0], sr[0])                                                # intrinsics
    assert np.abs(so[1] - sr[1]).max() <= 1e-15                                        # frame states (double)
    assert np.array_equal(so[2], sr[2])                                                # idepths (float)
    assert np.array_equal(rm, removed)                                                 # residuals dropped by linearizeAll(true)
    assert np.array_equal(O.frame_energy_th(), R.frame_energy_th())
    live = removed == 0
    a, b = O.residual_state(), R.residual_state()
    for k in a:
        assert np.array_equal(a[k][live], b[k][live]), k
    assert np.float32(np.sqrt(np.float32(eo / R.resInA()))) == np.float32(rmse)        # the function's return value
    assert np.abs(O.evalPT(W.nF - 1)[0] - R.evalPT(W.nF - 1)[0]).max() <= 1e-15         # setEvalPT of the newest frame
    prb, png = R.point_stats()
    assert np.array_equal(rb, prb) and np.array_equal(ng, png)                          # maxRelBaseline / numGoodResiduals bookkeeping


@needs_ref
def test_full_optimize_default_order_same_decisions(orc):
    W, O, R, tro, (eo, rb, ng, rm), (rmse, steps, removed, log) = _optimize_both(WINDOWS[0])
    assert [s[0] for s in steps] == [bool(a) for a in tro[:, 2]]
    so, sr = O.state(), R.state()
    assert np.abs(so[1] - sr[1]).max() <= 1e-12 and np.abs(so[2] - sr[2]).max() <= 1e-6 * np.abs(sr[2]).max()
    assert np.array_equal(rm, removed)


@needs_ref
def test_marginalisation(orc, eigen_order):
    cfg = dict(w=320, h=160, nF=5, pts_per_kf=120, seed=4, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
    W, O, R = _pair(cfg)
    mask = (np.random.default_rng(0).random(W.nP) < 0.3).astype(np.uint8)
    drop = ((np.random.default_rng(1).random(W.nP) < 0.1) & (mask == 0)).astype(np.uint8)
    for E in (O, R):
        E.linearizeAll(); E.applyRes()
        E.fixLinearization(mask)                                   # EFResidual::fixLinearizationF
    zo, lo = O.res_toZero()
    zr, lr = R.res_toZero()
    assert np.array_equal(lo, lr) and np.array_equal(zo, zr)
    O.marginalizePoints(mask, drop); R.marginalizePoints(mask, drop)     # addPoint<2>, SC addPoint(shiftPriorToZero = false), dropPointsF
    Ho, bo = O.marg_prior()
    Hr, br = R.marg_prior()
    assert _rel(Ho, Hr) <= 1e-12 and _rel(bo, br) <= 1e-12
    for idx in (0, 2, W.nF - 1):
        a, b = O.marginalizeFrame(idx), R.marginalizeFrame(idx)   # EnergyFunctional::marginalizeFrame
        assert _rel(a[0], b[0]) <= 1e-9 and _rel(a[1], b[1]) <= 1e-9
    # the window keeps working after the points are gone
    for E in (O, R):
        E.compute_nullspaces(); E.solveSystemF(0, 0.1)
    assert O.resInA() == R.resInA()
    so, sr = O.system(), R.system()
    # removePoint (EnergyFunctional.cpp:599-615) moves the host's last point into the freed slot, so the reference now accumulates the
    # surviving points in another order than the oracle (which keeps the input order): float sums agree to float rounding, not bit for bit
    assert _rel(so["HFinal"], sr["HFinal"]) <= 1e-9 and _rel(so["bFinal"], sr["bFinal"]) <= 1e-6
    assert _rel(so["x"], sr["x"]) <= 1e-4


@needs_ref
def test_write_or_check_golden(orc, eigen_order):
    """the fixture holds the REFERENCE's outputs for WINDOWS[0]; with the library present it must be reproducible"""
    from tools.gen_ref_pin_golden import backend_reference_outputs
    ref = backend_reference_outputs(WINDOWS[0])
    g = np.load(GOLD)
    for k in ref:
        assert np.array_equal(np.asarray(ref[k]), g[k], equal_nan=True), k


def test_oracle_against_reference_fixture(orc, eigen_order):
    """runs everywhere: the oracle against the reference outputs stored in tests/golden/ref_pin_backend.npz"""
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    g = np.load(GOLD)
    W = syn.make_window(**WINDOWS[0])
    O = OracleEF(W.w, W.h).load(W)
    O.compute_nullspaces()
    assert O.linearizeAll() == float(g["lin_energy"])
    assert np.array_equal(O.residual_J(0), g["J_new"])
    st = O.residual_state()
    assert np.array_equal(st["new_state"], g["new_state"]) and np.array_equal(st["new_energy"], g["new_energy"])
    O.applyRes()
    O.solveSystemF(0, 0.1)
    assert np.array_equal(O.top_acc(), g["top_acc"])
    for a, k in zip(O.sc_acc(), ("accE", "accEB", "accD", "Hcc", "bc")):
        assert np.array_equal(a, g[k])
    assert np.array_equal(O.points(), g["points"])
    assert _rel(O.system()["x"], g["x"]) <= 1e-9
    O2 = OracleEF(W.w, W.h).load(W)
    O2.compute_nullspaces()
    tr = O2.optimize(6)
    e, rb, ng, rm = O2.optimize_finish()
    assert [bool(a) for a in tr[:, 2]] == [bool(a) for a in g["opt_accept"]]
    s = O2.state()
    assert np.abs(s[1] - g["opt_state"]).max() <= 1e-15 and np.array_equal(s[2], g["opt_idepth"])
    assert np.array_equal(rm, g["opt_removed"])
