"""Known-answer tests pinning the CPU oracle's restatement of CoarseTracker::structPoseEstimation (SURVEY.md 8f-1;
CoarseTracker.cpp:840-1007).  The reference ships no tests for it, so these are the pins: an independent float64 numpy mirror of
calcHandb / calculateRes, a finite-difference check of the analytic Jacobian, and the documented control-flow quirks.  CPU only."""
import numpy as np
import pytest

from common import rel_err


def _mirror(orc, P, worldToCur7):
    """float64 numpy mirror of calcHandb + calculateRes at one pose (independent of the oracle's C++)."""
    from sdv_loam_amd import synthetic as syn
    fx, fy, cx, cy = (np.float64(np.float32(P.calib[k])) for k in ("fx", "fy", "cx", "cy"))
    R, t = syn.quat_to_R(np.asarray(worldToCur7[:4])), np.asarray(worldToCur7[4:])
    H = np.zeros((6, 6))
    b = np.zeros(6)
    e = 0.0
    num = 0
    for i in range(P.n):
        hp = P.host_poses7[P.host_idx[i]]
        X = syn.quat_to_R(hp[:4]) @ (np.array([(np.float64(P.u[i]) - cx) / fx, (np.float64(P.v[i]) - cy) / fy, 1.0])
                                     / np.float64(P.idepth[i])) + hp[4:]
        x, y, z = R @ X + t
        Ku, Kv = fx * x / z + cx, fy * y / z + cy
        if not (Ku > 1.1 and Kv > 1.1 and Ku < P.w - 3 and Kv < P.h - 3):
            continue
        ox, oy = np.float64(np.float32(P.obs[i, 0])), np.float64(np.float32(P.obs[i, 1]))
        e += (Ku - ox) ** 2 + (Kv - oy) ** 2
        num += 1
        jx = np.array([1 / z, 0, -x / z ** 2, -x * y / z ** 2, 1 - x * x / z ** 2, -y / z])
        jy = np.array([0, 1 / z, -y / z ** 2, -(1 - y * y / z ** 2), x * y / z ** 2, x / z])
        r = np.array([(Ku - ox) / fx, (Kv - oy) / fy])
        nrm2 = r @ r
        bsq = 4.6851 ** 2
        wgt = (1 - nrm2 / bsq) ** 2 if nrm2 <= bsq else 0.0
        J = np.stack([jx, jy])
        H += J.T @ J * wgt
        b += J.T @ r * wgt
    return H, b, e, num


def _problem(seed=0, **kw):
    from sdv_loam_amd import synthetic as syn
    return syn.make_struct_problem(n=300, seed=seed, **kw)


def _oracle(orc, P):
    T = orc.OracleTracker(P.w, P.h, 4)
    T.makeK(**P.calib)
    return T


def test_res_hb_matches_numpy_mirror(orc):
    P = _problem(0)
    T = _oracle(orc, P)
    w2c = orc.se3_inverse(P.init_curToWorld7)
    H, b, e, n = T.structResHb(w2c, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    Hm, bm, em, nm = _mirror(orc, P, w2c)
    assert n == nm and 0 < n < P.n                    # some matches fail the image-bounds test
    assert rel_err(H, Hm) < 2e-5 and rel_err(b, bm) < 2e-4 and abs(e - em) / em < 1e-4
    assert np.array_equal(H, H.T)


def test_jacobian_vs_true_derivative(orc):
    """calcHandb's Jacobian (:915-927) is the derivative of the unit-plane projection w.r.t. a LEFT se(3) increment EXCEPT for
    two entries: d_xi_x[4] = 1 + x*(-x/z^2) = 1 - u^2 and d_xi_y[3] = -(1 - v^2), where the true derivatives are 1 + u^2 and
    -(1 + v^2) (DSO's own tracker has the + sign, CoarseTracker.cpp:457,460).  The restatement keeps the reference's form; this test
    documents the deviation with finite differences (numpy only) and checks that the oracle's step is still a descent direction."""
    from sdv_loam_amd import synthetic as syn
    rng = np.random.default_rng(5)
    for _ in range(20):
        X = np.array([rng.uniform(-8, 8), rng.uniform(-3, 3), rng.uniform(4, 30)])
        x, y, z = X
        u, v = x / z, y / z
        jx_ref = np.array([1 / z, 0, -x / z ** 2, -x * y / z ** 2, 1 - x * x / z ** 2, -y / z])
        jy_ref = np.array([0, 1 / z, -y / z ** 2, -(1 - y * y / z ** 2), x * y / z ** 2, x / z])
        fd = np.zeros((2, 6))
        for k in range(6):
            d = np.zeros(6)
            d[k] = 1e-6
            Tp, Tm = syn.se3_exp_np(d), syn.se3_exp_np(-d)
            Xp = syn.quat_to_R(Tp[:4]) @ X + Tp[4:]
            Xm = syn.quat_to_R(Tm[:4]) @ X + Tm[4:]
            fd[:, k] = (Xp[:2] / Xp[2] - Xm[:2] / Xm[2]) / 2e-6
        dx, dy = jx_ref - fd[0], jy_ref - fd[1]
        assert np.allclose(np.delete(dx, 4), 0, atol=1e-7) and np.allclose(np.delete(dy, 3), 0, atol=1e-7)
        assert dx[4] == pytest.approx(-2 * u * u, abs=1e-7) and dy[3] == pytest.approx(2 * v * v, abs=1e-7)
    # the (slightly wrong) Gauss-Newton step still reduces the error of a noise-free problem substantially
    P = _problem(1, noise_px=0.0, outlier_frac=0.0, pose_err=(0.01, 0.001))
    T = _oracle(orc, P)
    w2c = orc.se3_inverse(P.init_curToWorld7)
    H, b, e0, n = T.structResHb(w2c, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    new = orc.se3_mul(orc.se3_exp(np.linalg.solve(H, -b)), w2c)
    _, _, e1, n1 = T.structResHb(new, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    assert e1 / n1 < 0.2 * e0 / n


def test_lm_trace_follows_the_reference_control_flow(orc):
    P = _problem(2)
    T = _oracle(orc, P)
    pose, tr, fr = T.structPoseEstimation(P.init_curToWorld7, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    assert 1 <= len(tr) <= 10
    lam = 0.01
    res_old = tr[0, 2]
    for k, row in enumerate(tr):
        assert row[0] == k
        assert row[1] == pytest.approx(np.float32(lam), rel=1e-6)
        assert row[2] == pytest.approx(res_old, rel=1e-6)
        accept = row[3] < row[2]
        assert bool(row[4]) == accept
        if accept:
            res_old = row[3]
            lam *= 0.5
        else:
            lam = max(lam * 4, 0.001)
        if k < len(tr) - 1:
            assert row[13] > 1e-5                      # the loop only continues while |inc| > 1e-5
    assert tr[-1, 13] <= 1e-5 or len(tr) == 10
    assert fr == pytest.approx(res_old, rel=1e-6)
    assert tr[0, 4] == 1                               # first step from a 5 cm / 0.2 deg error is accepted
    # accepted steps move the pose towards the ground truth
    e0 = np.linalg.norm(orc.se3_log(orc.se3_mul(orc.se3_inverse(P.gt_curToWorld7), P.init_curToWorld7)))
    e1 = np.linalg.norm(orc.se3_log(orc.se3_mul(orc.se3_inverse(P.gt_curToWorld7), pose)))
    assert e1 < 0.5 * e0


def test_stale_linearisation_point_quirk(orc):
    """After the first accepted step the reference rebuilds H,b at the pose BEFORE the step (:983), so the second trial increment
    equals the first one re-damped (same H,b, smaller lambda) -- not a fresh Gauss-Newton step."""
    P = _problem(3, noise_px=0.0, outlier_frac=0.0, pose_err=(0.05, 0.004))
    T = _oracle(orc, P)
    w2c = orc.se3_inverse(P.init_curToWorld7)
    H, b, _, _ = T.structResHb(w2c, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    _, tr, _ = T.structPoseEstimation(P.init_curToWorld7, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    assert tr[0, 4] == 1 and len(tr) >= 2
    H1 = H.copy()
    H1[np.diag_indices(6)] *= np.float64(np.float32(1) + np.float32(0.01))
    assert rel_err(tr[0, 5:11], np.linalg.solve(H1, -b)) < 1e-6
    H2 = H.copy()                                                    # rebuilt undamped at the OLD pose, then damped with 0.005
    H2[np.diag_indices(6)] *= np.float64(np.float32(1) + np.float32(0.005))
    assert rel_err(tr[1, 5:11], np.linalg.solve(H2, -b)) < 1e-6


def test_cumulative_damping_quirk(orc):
    """Rejected steps keep multiplying the SAME H's diagonal (:959): after rejections with lambda l1, l2 the diagonal carries
    (1+l1)(1+l2), not (1+l2)."""
    P = _problem(3, noise_px=0.0, outlier_frac=0.0, pose_err=(0.05, 0.004))
    T = _oracle(orc, P)
    w2c = orc.se3_inverse(P.init_curToWorld7)
    H, b, _, _ = T.structResHb(w2c, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    _, tr, _ = T.structPoseEstimation(P.init_curToWorld7, P.u, P.v, P.idepth, P.host_idx, P.host_poses7, P.obs)
    assert len(tr) >= 4 and tr[0, 4] == 1 and tr[1, 4] == 0 and tr[2, 4] == 0
    f = np.float64(1.0)
    for k in (1, 2, 3):
        f *= np.float64(np.float32(1) + np.float32(tr[k, 1]))
        Hk = H.copy()
        Hk[np.diag_indices(6)] *= f
        assert rel_err(tr[k, 5:11], np.linalg.solve(Hk, -b)) < 1e-6


def test_no_inbounds_match(orc):
    """num == 0: resOld = 0/0 = NaN, every trial is rejected (NaN comparisons), the pose is returned unchanged."""
    P = _problem(4)
    T = _oracle(orc, P)
    u = np.full(P.n, 2.0, np.float32)                 # all points far left and close: they leave the image
    idepth = np.full(P.n, 0.5, np.float32)
    far = P.init_curToWorld7.copy()
    far[4:] += [500.0, 0, 0]
    pose, tr, fr = T.structPoseEstimation(far, u, P.v, idepth, P.host_idx, P.host_poses7, P.obs)
    assert np.array_equal(pose, far)
    assert np.all(tr[:, 4] == 0) and np.all(tr[:, 11] == 0)
    assert np.isnan(fr)
