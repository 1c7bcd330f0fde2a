"""Analytic known-answer tests pinning the back-end oracle.  CPU only."""
import numpy as np
import pytest

from backend_model import T4, dense_normal_equations, project, relative_jacobians_fd
from common import rel_err


@pytest.fixture(scope="module")
def small_window():
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=640, h=240, nF=4, pts_per_kf=150, seed=1, calib=dict(fx=400., fy=410., cx=319.5, cy=119.5))


@pytest.fixture()
def ef(orc, small_window):
    from oracle.backend import OracleEF
    return OracleEF(small_window.w, small_window.h).load(small_window)


def test_states_and_energy(ef, small_window):
    W = small_window
    e = ef.linearizeAll()
    st = ef.residual_state()
    assert e > 0 and np.isfinite(e)
    n_in, n_oob, n_out = [(st["new_state"] == k).sum() for k in range(3)]
    assert n_in > 0.6 * W.nR and n_in + n_oob + n_out == W.nR
    assert np.all(st["energy_with_outlier"][st["new_state"] == 1] == -1)      # OOB: state_NewEnergyWithOutlier stays -1
    out = st["new_state"] == 2
    assert np.all(st["new_energy"][out] == 512.0)                              # clamped to max(frameEnergyTH)
    ef.applyRes()
    st2 = ef.residual_state()
    assert np.array_equal(st2["state"], st["new_state"]) and np.array_equal(st2["active"], (st["new_state"] == 0).astype(np.uint8))
    # residuals without a matcher or that start OOB stay OOB and are never activated
    W2 = small_window
    r_has = W2.r_hasMatcher.copy(); r_has[::7] = 0
    r_st = W2.r_state.copy(); r_st[3::11] = 1
    from oracle.backend import OracleEF
    import copy
    W3 = copy.copy(W2); W3.r_hasMatcher = r_has; W3.r_state = r_st
    E3 = OracleEF(W3.w, W3.h).load(W3)
    E3.linearizeAll(); E3.applyRes()
    s3 = E3.residual_state()
    assert np.all(s3["state"][::7] == 1) and np.all(s3["state"][3::11] == 1) and not s3["active"][::7].any()


def test_relative_jacobians_by_finite_differences(orc, ef, small_window):
    W = small_window
    ef.linearizeAll()
    J = ef.residual_J(0)
    st = ef.residual_state()
    Kv = W.value_scaled
    rng = np.random.default_rng(0)
    cand = np.where(st["new_state"] != 1)[0]
    for r in rng.choice(cand, 40, replace=False):
        p = W.r_point[r]; h = W.host[p]; t = W.r_target[r]
        T0 = T4(W.evalPT[t]) @ np.linalg.inv(T4(W.evalPT[h]))
        Jfd = relative_jacobians_fd(Kv, T0, W.u[p], W.v[p], W.idepth_zero[p])
        res = project(Kv, T0, W.u[p], W.v[p], W.idepth_zero[p]) - W.r_matcher[r]
        nrm = np.linalg.norm(res)
        hw = 1.0 if nrm < 6 else np.sqrt(6 / nrm)
        Jo = np.concatenate([J[r, 2:8], J[r, 14:18], J[r, 22:23]]), np.concatenate([J[r, 8:14], J[r, 18:22], J[r, 23:24]])
        for a in range(2):
            assert np.allclose(Jo[a], hw * Jfd[a], rtol=2e-3, atol=2e-3 * np.abs(Jfd[a]).max()), (r, a, Jo[a], hw * Jfd[a])
        assert np.allclose(J[r, 0:2], hw * res, rtol=1e-4, atol=1e-3)


def test_schur_solution_matches_dense_solve(orc, ef, small_window):
    """x (frames+calib) and the per-point idepth steps from accumulate -> Schur -> LDLT -> resubstitute must equal the
    direct float64 solution of the full normal equations (lambda = 0, no marginalisation prior)."""
    W = small_window
    ef.L.orc_ef_set_marg_prior(ef.h_, np.zeros(ef.dim * ef.dim), np.zeros(ef.dim))
    ef.linearizeAll(); ef.applyRes()
    ef.solveSystemF(0, 0.0)
    s = ef.system()
    J = ef.residual_J(1)
    active = ef.residual_state()["active"].astype(bool)
    H, b, n, free, pidx = dense_normal_equations(W, ef, J, active)
    # priors: calib 5e9 (b += prior*cDelta), first frame 1e10/1e11 (b += prior*delta_prior), points hosted in KF0 50^2
    cD = W.value_minus_value_zero.astype(np.float32).astype(np.float64)
    H[np.arange(4), np.arange(4)] += 5e9
    b[:4] += 5e9 * cD
    H[np.arange(4, 7), np.arange(4, 7)] += 1e10
    H[np.arange(7, 10), np.arange(7, 10)] += 1e11
    b[4:10] += np.array([1e10] * 3 + [1e11] * 3) * W.state[0, :6]
    for p in free:
        if W.hasDepthPrior[p]:
            H[n + pidx[p], n + pidx[p]] += 2500.0
    # points without any active residual are not unknowns of the reference's system (step = 0)
    npts_active = np.zeros(W.nP, int)
    np.add.at(npts_active, W.r_point[active], 1)
    keep = np.concatenate([np.arange(n), n + pidx[[p for p in free if npts_active[p] > 0]]])
    Hk, bk = H[np.ix_(keep, keep)], b[keep]
    # floor on Hdd like AccumulatedSCHessian.cpp:26
    sol = np.linalg.solve(Hk, bk)
    x_dense = sol[:n]
    assert rel_err(s["x"][4:], x_dense[4:]) < 2e-3
    assert np.allclose(s["x"][:4], x_dense[:4], rtol=2e-3, atol=1e-9)
    steps = ef.points()[:, 8].astype(np.float64)
    dense_steps = np.zeros(W.nP)
    kept_pts = [p for p in free if npts_active[p] > 0]
    dense_steps[kept_pts] = -sol[n:]
    assert rel_err(steps, dense_steps) < 5e-3
    assert np.all(steps[W.isFromSensor == 1] == 0)
    fs, cs = ef.frame_steps()
    assert np.array_equal(fs.reshape(-1), -s["x"][4:]) and np.array_equal(cs, -s["x"][:4])


def test_top_accumulators_match_float64_sums(orc, ef, small_window):
    W = small_window
    ef.linearizeAll(); ef.applyRes(); ef.solveSystemF(0, 0.1)
    J = ef.residual_J(1).astype(np.float64)
    active = ef.residual_state()["active"].astype(bool)
    acc = ef.top_acc()
    nF = W.nF
    ref = np.zeros((nF * nF, 13, 13))
    for r in np.where(active)[0]:
        h = W.host[W.r_point[r]]; t = W.r_target[r]
        x = np.concatenate([J[r, 14:18], J[r, 2:8]]); y = np.concatenate([J[r, 18:22], J[r, 8:14]])
        A = ref[h + t * nF]
        A[:10, :10] += np.outer(x, x) + np.outer(y, y)
        A[:10, 12] += x * J[r, 0] + y * J[r, 1]
        A[12, :10] = A[:10, 12]
        A[12, 12] += J[r, 0] ** 2 + J[r, 1] ** 2
    for k in range(nF * nF):
        if np.abs(ref[k]).max() > 0:
            assert rel_err(acc[k], ref[k]) < 1e-5
        else:
            assert np.all(acc[k] == 0)
    assert ef.resInA() == active.sum()
    # per-point sums
    pts = ef.points()
    Hdd = np.zeros(W.nP); bd = np.zeros(W.nP); Hcd = np.zeros((W.nP, 4))
    for r in np.where(active)[0]:
        p = W.r_point[r]
        Hdd[p] += J[r, 22] ** 2 + J[r, 23] ** 2
        bd[p] += J[r, 0] * J[r, 22] + J[r, 1] * J[r, 23]
        Hcd[p] += J[r, 14:18] * J[r, 22] + J[r, 18:22] * J[r, 23]
    assert rel_err(pts[:, 0], Hdd) < 1e-6 and rel_err(pts[:, 1], bd) < 1e-5 and rel_err(pts[:, 2:6], Hcd) < 1e-5


def test_system_symmetry_lambda_and_orthogonalize(orc, ef, small_window):
    W = small_window
    ef.linearizeAll(); ef.applyRes()
    ef.solveSystemF(0, 0.1)
    s = ef.system()
    assert np.allclose(s["HA"], s["HA"].T, rtol=0, atol=1e-9 * np.abs(s["HA"]).max())
    assert np.allclose(s["Hsc"], s["Hsc"].T, rtol=1e-5, atol=1e-6 * np.abs(s["Hsc"]).max())
    n = ef.dim
    Hd = s["HFinal"].copy(); Hd[np.arange(n), np.arange(n)] *= 1.1
    assert rel_err(s["x"], np.linalg.solve(Hd, s["bFinal"])) < 1e-6
    # iteration >= 2: x is projected off the given null-space vectors (EnergyFunctional.cpp:746-750, 615-648)
    rng = np.random.default_rng(3)
    ns = rng.normal(size=(7, n))
    ef.L.orc_ef_set_nullspaces(ef.h_, 7, np.ascontiguousarray(ns).reshape(-1))
    ef.solveSystemF(2, 0.1)
    x2 = ef.system()["x"]
    Q, _ = np.linalg.qr(ns.T)
    assert rel_err(x2, s["x"] - Q @ (Q.T @ s["x"])) < 1e-9


def test_threaded_mode_matches_single_thread(orc):
    """oracle multi-thread timing mode (the reference's multiThreading=true IndexThreadReduce paths): per-worker accumulators summed
    in double at the stitch -> same accept/reject sequence, energies and increments to float-summation-order accuracy."""
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=320, h=160, nF=5, pts_per_kf=150, seed=11, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
    out = {}
    for T in (1, 3):
        O = OracleEF(W.w, W.h).load(W)
        O.set_threads(T)
        tr = O.optimize(5)
        out[T] = (tr, O.system()["HFinal"], O.state())
    a, b = out[1][0], out[3][0]
    assert len(a) == len(b) and (a[:, 2] == b[:, 2]).all()                                   # accept / reject sequence
    assert np.abs(a[:, 3] - b[:, 3]).max() <= 1e-5 * np.abs(a[:, 3]).max()                    # energies
    assert np.abs(a[:, 7:] - b[:, 7:]).max() <= 1e-4 * np.abs(a[:, 7:]).max()                 # increments (north_star tolerance)
    Ha, Hb = out[1][1], out[3][1]
    assert np.linalg.norm(Ha - Hb) <= 1e-5 * np.linalg.norm(Ha)


def _th_mirror(energy_wo, target, lin, nF):
    """numpy mirror of FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:63-97), float32 like the reference."""
    v = energy_wo[(target == nF - 1) & (lin == 0) & (energy_wo >= 0)].astype(np.float32)
    if v.size == 0:
        return np.float32(12 * 12 * 8)
    k = int(np.float32(0.7) * np.float32(v.size))
    nth = np.sqrt(np.partition(v, k)[k], dtype=np.float32)
    th = nth * np.float32(1.5)
    th = np.float32(26.0) * np.float32(0.5) + th * np.float32(0.5)
    return np.float32(th * th)


def test_set_new_frame_energy_th(orc, small_window):
    """linearizeAll ends with setNewFrameEnergyTH: the newest frame's threshold is the numpy-partition quantile formula, the other
    frames keep theirs, and the NEXT linearise classifies IN / OUTLIER with max(host TH, target TH) (Residuals.cpp:212-214)."""
    import copy
    from oracle.backend import OracleEF
    W = copy.copy(small_window)
    W.frameEnergyTH = np.array([150, 170, 190, 512], np.float32)   # older frames below the newest's: its threshold decides (max of the two)
    E = OracleEF(W.w, W.h).load(W)
    th0 = E.frame_energy_th()
    assert np.array_equal(th0, W.frameEnergyTH)
    E.linearizeAll()
    st = E.residual_state()
    th1 = E.frame_energy_th()
    want = _th_mirror(st["energy_with_outlier"], W.r_target, W.r_isLinearized, W.nF)
    assert th1[-1] == want and np.array_equal(th1[:-1], th0[:-1]) and th1[-1] != 512.0
    # same state again: same energies -> same quantile, but the classification now uses the new threshold
    E.linearizeAll()
    st2 = E.residual_state()
    assert E.frame_energy_th()[-1] == want
    assert np.array_equal(st2["energy_with_outlier"], st["energy_with_outlier"])
    h = W.host[W.r_point]
    thr = np.maximum(th1[h], th1[W.r_target]).astype(np.float64)
    live = st2["new_state"] != 1
    over = st2["energy_with_outlier"] > thr
    assert np.all(st2["new_state"][live & over] == 2) and np.all(st2["new_energy"][live & over] == thr[live & over])
    assert np.all(st2["new_energy"][live & (st2["new_state"] == 0)] == st2["energy_with_outlier"][live & (st2["new_state"] == 0)])
    moved = (st2["new_state"] != st["new_state"]).sum()
    assert moved > 0      # the window is built so that the threshold matters
    # no candidate at all -> 12*12*patternNum
    W2 = copy.copy(W)
    W2.r_hasMatcher = W.r_hasMatcher.copy(); W2.r_hasMatcher[W.r_target == W.nF - 1] = 0     # -> OOB, energy -1
    E2 = OracleEF(W2.w, W2.h).load(W2)
    E2.linearizeAll()
    assert E2.frame_energy_th()[-1] == 12 * 12 * 8


def test_optimize_trace_carries_threshold_and_finish(orc, small_window):
    import copy
    from oracle.backend import OracleEF
    W = copy.copy(small_window)
    W.frameEnergyTH = np.array([150, 170, 190, 300], np.float32)
    E = OracleEF(W.w, W.h).load(W)
    tr = E.optimize(6)
    n = E.dim
    assert tr.shape[1] == 8 + n and np.all(tr[:, 7 + n] > 0) and len(set(tr[:, 7 + n])) > 1
    st_before = E.state()[1]
    e, rb, ng, rm = E.optimize_finish()
    assert np.isfinite(e) and e > 0
    p7, z = E.evalPT(W.nF - 1)
    # setEvalPT: the state is zero except the affine part, which is kept; state_zero == state
    st = E.state()[1]
    assert np.all(st[-1, :6] == 0) and np.array_equal(st[-1, 6:8], st_before[-1, 6:8]) and np.array_equal(z, st[-1])
    rs = E.residual_state()
    # toRemove == not active after applyRes(true); surviving residuals are counted per point
    assert np.array_equal(rm.astype(bool), rs["active"] == 0)
    cnt = np.bincount(W.r_point[rs["active"] == 1], minlength=W.nP)
    assert np.array_equal(cnt, ng)
    assert np.all(rb[ng > 0] > 0) and np.all(rb[ng == 0] == 0)
    want = _th_mirror(rs["energy_with_outlier"], W.r_target, W.r_isLinearized, W.nF)
    assert E.frame_energy_th()[-1] == want


def test_nullspaces_known_answers():
    """FrameHessian::setStateZero's central differences have closed forms: T exp(eps) T^-1 = exp(Ad_T eps), so the pose columns are the
    columns of Ad_T; scaling the translation gives T_s T^-1 = (I, (s - 1) t), so the scale vector is [(1.00001 - 1/1.00001) / 2e-3 * t, 0].
    getNullspaces then divides the translation part by SCALE_XI_TRANS (x2) and the rotation part by SCALE_XI_ROT (x1)."""
    import oracle
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=320, h=160, nF=4, pts_per_kf=40, seed=5, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
    O = OracleEF(W.w, W.h).load(W)
    ns = O.compute_nullspaces()
    assert ns.shape == (7, 4 + 6 * W.nF) and np.all(ns[:, :4] == 0)
    sinv = np.array([2.0, 2.0, 2.0, 1.0, 1.0, 1.0])
    for h in range(W.nF):
        T = np.asarray(W.evalPT[h], np.float64)
        Ad = oracle.se3_adj(T)
        got = ns[:6, 4 + 6 * h:10 + 6 * h]                    # [direction i][row r]
        assert np.allclose(got, (Ad * sinv[:, None]).T, rtol=1e-6, atol=1e-9)
        t = T[4:7]
        k = (1.00001 - 1 / 1.00001) / 2e-3
        assert np.allclose(ns[6, 4 + 6 * h:7 + 6 * h], 2.0 * k * t, rtol=1e-5, atol=1e-12)
        assert np.allclose(ns[6, 7 + 6 * h:10 + 6 * h], 0, atol=1e-9)


def test_nullspaces_product_matches_oracle():
    """sdvgn_ef_compute_nullspaces (host arithmetic of the product, exercised on a host-only handle) against the oracle, then the
    projected solve uses them: x is orthogonal to the installed gauge directions up to the reference's 0.5 (N Npi^T + Npi N^T) form."""
    from oracle.backend import OracleEF
    from sdv_loam_amd import backend_api, synthetic as syn
    W = syn.make_window(w=320, h=160, nF=5, pts_per_kf=40, seed=6, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))
    O = OracleEF(W.w, W.h).load(W)
    no = O.compute_nullspaces()
    G = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP, device=-1)
    c = np.ascontiguousarray
    G.nF = W.nF
    assert G.L.sdvgn_ef_set_frames(G.h_, W.nF, c(W.evalPT, np.float64).reshape(-1), c(W.state, np.float64).reshape(-1), c(W.state_zero, np.float64).reshape(-1),
                                   c(W.frameID, np.int32), c(W.ab_exposure, np.float32), c(W.frameEnergyTH, np.float32)) == 0
    ng = G.compute_nullspaces()
    assert ng.shape == no.shape and np.allclose(ng, no, rtol=1e-12, atol=1e-13)
