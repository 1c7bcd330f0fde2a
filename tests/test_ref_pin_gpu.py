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
