"""Oracle KATs for the marginalisation step around FullSystem::optimize: EFResidual::fixLinearizationF
(EnergyFunctionalStructs.cpp:45-55), EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:514-576, addPoint<2>) and
marginalizeFrame (:434-512).  The reference holds no vectors for these (the pin against its own code: tests/test_ref_pin_backend.py::test_marginalisation): numpy mirrors + properties."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def setup(orc):
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=320, h=160, nF=5, pts_per_kf=120, seed=4, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5))

    def fresh():
        O = OracleEF(W.w, W.h).load(W)
        O.linearizeAll(); O.applyRes()
        return O
    return W, fresh


def test_fix_linearization_numpy_mirror(setup):
    """res_toZeroF = resF - (Jpdxi . adHTdeltaF + Jpdc . cDeltaF + Jpdd * deltaF) for the active residuals of the flagged points only."""
    W, fresh = setup
    O = fresh()
    mask = (np.random.default_rng(0).random(W.nP) < 0.3).astype(np.uint8)
    J = O.residual_J(1)                                    # EFResidual::J, 24 floats: resF(2) Jpdxi[0](6) Jpdxi[1](6) Jpdc[0](4) Jpdc[1](4) Jpdd(2)
    st = O.residual_state()
    O.fixLinearization(mask)
    r2z, lin = O.res_toZero()
    act = st["active"].astype(bool)
    sel = mask[W.r_point].astype(bool) & act
    assert np.array_equal(lin.astype(bool), sel) and sel.sum() > 50
    # deltas: frames state - state_zero through the adjoints -> use the oracle's own linear model: mode-1 resApprox must reproduce resF
    # (res_toZero + J . delta == resF up to float rounding), which pins the sign and the three terms together
    dp = O.adHTdeltaF()                                    # [h + nF*t][6]
    cd = np.asarray(W.value_minus_value_zero, np.float32)
    dd = (W.idepth - W.idepth_zero).astype(np.float32)
    k = W.host[W.r_point] + W.nF * W.r_target
    jx = np.einsum("ij,ij->i", J[:, 2:8], dp[k]) + J[:, 14:18] @ cd + J[:, 22] * dd[W.r_point]
    jy = np.einsum("ij,ij->i", J[:, 8:14], dp[k]) + J[:, 18:22] @ cd + J[:, 23] * dd[W.r_point]
    exp = np.stack([J[:, 0] - jx, J[:, 1] - jy], 1)
    assert np.allclose(r2z[sel], exp[sel], rtol=1e-5, atol=1e-5)
    assert not r2z[~sel].any()


def test_marginalize_points_linearity_and_removal(setup):
    """HM/bM gain 0.25 (M - Msc) of the flagged points: two batches add up to the joint batch; the result is symmetric to rounding;
    marginalised points no longer take part in the next solve (their residuals are gone), dropped points likewise but add nothing."""
    W, fresh = setup
    rng = np.random.default_rng(1)
    m1 = (rng.random(W.nP) < 0.15).astype(np.uint8)
    m2 = ((rng.random(W.nP) < 0.15) & (m1 == 0)).astype(np.uint8)
    A = fresh(); A.fixLinearization(m1 | m2); A.marginalizePoints(m1 | m2)
    B = fresh(); B.fixLinearization(m1 | m2); B.marginalizePoints(m1); B.marginalizePoints(m2)
    Ha, ba = A.marg_prior(); Hb, bb = B.marg_prior()
    H0, b0 = fresh().marg_prior()
    assert np.linalg.norm(Ha - H0) > 1e-3 * max(1.0, np.linalg.norm(H0))
    # (the accumulators are float32 with tiered shift-up, so a different batching differs in the last float bits)
    assert np.linalg.norm(Ha - Hb) <= 1e-6 * np.linalg.norm(Ha) and np.linalg.norm(ba - bb) <= 1e-6 * np.linalg.norm(ba)
    assert np.abs(Ha - Ha.T).max() <= 1e-8 * np.abs(Ha).max()
    # dropping adds nothing to the prior
    D = fresh(); D.marginalizePoints(np.zeros(W.nP, np.uint8), m1)
    Hd, bd = D.marg_prior()
    assert np.array_equal(Hd, H0) and np.array_equal(bd, b0)
    # and both ways the points vanish from the active system: resInA drops by their active residuals
    st = fresh().residual_state()
    act = st["active"].astype(bool)
    n_gone = int((act & (m1[W.r_point] > 0)).sum())
    F = fresh(); F.solveSystemF(0, 0.1); n0 = F.resInA()
    D.solveSystemF(0, 0.1)
    assert D.resInA() == n0 - n_gone
    B2 = fresh(); B2.fixLinearization(m1); B2.marginalizePoints(m1); B2.solveSystemF(0, 0.1)
    assert B2.resInA() == n0 - n_gone


def test_marginalize_frame_numpy_mirror(setup):
    """Permutation to the end + prior + preconditioned Schur complement (EnergyFunctional.cpp:446-493) against numpy in double."""
    W, fresh = setup
    O = fresh()
    m = (np.random.default_rng(2).random(W.nP) < 0.3).astype(np.uint8)
    O.fixLinearization(m); O.marginalizePoints(m)           # a non-trivial HM / bM
    HM, bM = O.marg_prior()
    n = O.dim
    for idx in (0, 2, W.nF - 1):
        Ho, bo = O.marginalizeFrame(idx)
        blk = list(range(4 + 6 * idx, 4 + 6 * idx + 6))
        perm = [i for i in range(n) if i not in blk] + blk
        H = HM[np.ix_(perm, perm)].copy(); b = bM[perm].copy()
        prior, dprior = O.frame_prior(idx)
        H[-6:, -6:] += np.diag(prior); b[-6:] += prior * dprior
        S = np.sqrt(np.abs(np.diag(H)) + 10)
        Hs = H / S[:, None] / S[None, :]; bs = b / S
        hpi = np.linalg.inv(Hs[-6:, -6:])
        bli = Hs[-6:, :-6].T @ hpi
        Ht = Hs[:-6, :-6] - bli @ Hs[-6:, :-6]
        bt = bs[:-6] - bli @ bs[-6:]
        Ht = Ht * S[:-6, None] * S[None, :-6]; bt = bt * S[:-6]
        Ht = 0.5 * (Ht + Ht.T)
        assert Ho.shape == (n - 6, n - 6)
        assert np.linalg.norm(Ho - Ht) <= 1e-9 * np.linalg.norm(Ht) and np.linalg.norm(bo - bt) <= 1e-9 * max(1.0, np.linalg.norm(bt))
        assert np.array_equal(Ho, Ho.T)


def test_eigen_reduction_order_sensitivity(orc):
    """What the order Eigen adds the terms of its small dot products in can do to this path (oracle/README.md: the one thing the parity
    statement cannot settle without Eigen).  The restatement adds left to right; with the halving unroller of Eigen's scalar reductions
    (order 1) and with SSE packets (order 2) in calcLEnergyPt / fixLinearizationF / addPoint<1> / resubstituteFPt, a window with
    linearised residuals and a marginalisation prior takes the same accept / reject decisions and ends where the tolerance of the
    contract (1e-4 on the increments) cannot tell the runs apart."""
    from oracle.backend import OracleEF
    from sdv_loam_amd import synthetic as syn
    W = syn.make_window(w=320, h=160, nF=5, pts_per_kf=120, seed=5, calib=dict(fx=200., fy=205., cx=159.5, cy=79.5),
                        state_sigma=1e-3, idepth_sigma=0.01)
    mask = (np.random.default_rng(3).random(W.nP) < 0.25).astype(np.uint8)
    L = orc.lib()
    runs = []
    try:
        for order in (0, 1, 2):
            L.orc_set_redux_order(order)
            O = OracleEF(W.w, W.h).load(W)
            O.linearizeAll(); O.applyRes()
            O.fixLinearization(mask)                     # linearised residuals: the L energy and addPoint<1> are live
            r2z, _ = O.res_toZero()
            tr = O.optimize(6, fixed_its=True)
            runs.append((tr, O.state(), r2z))
    finally:
        L.orc_set_redux_order(0)
    t0, s0, z0 = runs[0]
    assert (t0[:, 2] == 1).any() and (t0[:, 2] == 0).any()                              # accepted and rejected steps
    assert np.abs(t0[:, 4]).max() > 0                                                   # the L energy is not trivially zero
    for tr, st, r2z in runs[1:]:
        assert np.array_equal(tr[:, 2], t0[:, 2])                                       # same decisions
        assert np.allclose(tr[:, 3:6], t0[:, 3:6], rtol=1e-5, atol=1e-4)                # energies
        n = np.linalg.norm(t0[:, 7:59], axis=1)
        d = np.linalg.norm(tr[:, 7:59] - t0[:, 7:59], axis=1)
        assert (d <= 1e-4 * np.maximum(n, 1e-12)).all(), (d / np.maximum(n, 1e-12)).max()
        assert np.allclose(r2z, z0, rtol=1e-5, atol=1e-5)
        for a, b in zip(st, s0):
            assert np.allclose(a, b, rtol=1e-5, atol=1e-7)
    # and the orders ARE different arithmetic (otherwise this test shows nothing)
    assert any(not np.array_equal(r[2], z0) or not np.array_equal(r[0], t0) for r in runs[1:])
