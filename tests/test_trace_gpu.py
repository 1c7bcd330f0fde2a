"""GPU parity of sdvgn_tracker_trace_points (ImmaturePoint::traceOn, src/FullSystem/ImmaturePoint.cpp:47-353)
against the CPU oracle, through the C ABI.  One lane walks one epipolar segment in the reference's order: every output is expected
bit-identical (array_equal, NaN-aware)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)


def _window(seed=2, n=250, nF=4):
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=320, h=200, nF=nF, pts_per_kf=n, seed=seed, calib=CAL)


def _gpu(P):
    from sdv_loam_amd import api
    G = api.CoarseTracker(P.w, P.h, 3, max_points=64)
    G.makeK(**CAL)
    G.set_new_image(P.image, 1.0)
    G.traceSetPoints(P.u, P.v, P.energyTH, P.gradH, P.color, P.weights, P.host_idx)
    return G


def _same(a, b):
    for k in ("status", "idepth_min", "idepth_max", "quality", "lastTraceUV", "interval"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


@pytest.mark.parametrize("seed,pose_err", [(2, (0.0, 0.0)), (3, (0.02, 0.002)), (4, (0.1, 0.01))])
def test_trace_parity_first_and_second_frame(orc, seed, pose_err):
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    W = _window(seed)
    P = syn.make_trace_problem(W, target=2, pose_err=pose_err, seed=seed)
    G = _gpu(P)
    so = trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status)
    sg = G.tracePoints(P.KRKi, P.Kt, P.aff, P.idepth_min, P.idepth_max, P.quality, P.status)
    _same(sg, so)
    assert (so["status"] == syn.IPS_GOOD).sum() > 0.4 * P.n
    # second frame with the updated state (finite idepth_max branch, SKIPPED / BADCONDITION / OUTLIER->OOB transitions)
    P2 = syn.make_trace_problem(W, target=3, pose_err=pose_err, seed=seed + 1)
    common = np.nonzero(W.host < 2)[0]
    a = np.searchsorted(np.nonzero(W.host != 2)[0], common)
    b = np.searchsorted(np.nonzero(W.host != 3)[0], common)
    st = {k: getattr(P2, k).copy() for k in ("idepth_min", "idepth_max", "quality", "status")}
    for k in st:
        st[k][b] = so[k][a]
    G2 = _gpu(P2)
    so2 = trace_on(P2, P2.dI, st["idepth_min"], st["idepth_max"], st["quality"], st["status"])
    sg2 = G2.tracePoints(P2.KRKi, P2.Kt, P2.aff, st["idepth_min"], st["idepth_max"], st["quality"], st["status"])
    _same(sg2, so2)
    assert len(np.unique(so2["status"])) >= 3


def test_trace_parity_branches_and_affine(orc):
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    W = _window(5, n=300)
    P = syn.make_trace_problem(W, seed=5)
    P.aff[:, 0] = np.float32(0.9)
    P.aff[:, 1] = np.float32(4.0)                      # brightness transfer host -> new frame
    P.energyTH[::7] = np.nan                          # constructor saw a non-finite colour -> OUTLIER
    imin = P.idepth_min.copy()
    imax = P.idepth_max.copy()
    imin[1::5] = P.true_idepth[1::5] * np.float32(0.999)   # already certain -> SKIPPED
    imax[1::5] = P.true_idepth[1::5] * np.float32(1.001)
    imin[2::5] = P.true_idepth[2::5] * np.float32(0.9)     # wide finite interval -> searched
    imax[2::5] = P.true_idepth[2::5] * np.float32(1.1)
    status = P.status.copy()
    status[3::11] = syn.IPS_OOB                         # untouched
    status[4::11] = syn.IPS_OUTLIER                     # a second failure turns into OOB
    G = _gpu(P)
    so = trace_on(P, P.dI, imin, imax, P.quality, status)
    sg = G.tracePoints(P.KRKi, P.Kt, P.aff, imin, imax, P.quality, status)
    _same(sg, so)
    assert set(np.unique(so["status"])) >= {syn.IPS_GOOD, syn.IPS_OOB, syn.IPS_OUTLIER, syn.IPS_SKIPPED}


def test_trace_large_image_and_errors(orc):
    """KITTI-sized frame: maxPixSearch = 43.7 px -> up to 45 search steps (several 4-step batches) per point."""
    from oracle.trace import trace_on
    from sdv_loam_amd import api, synthetic as syn
    W = syn.make_window(w=1241, h=376, nF=3, pts_per_kf=400, seed=6, calib=syn.KITTI00)
    P = syn.make_trace_problem(W, seed=6)
    G = api.CoarseTracker(P.w, P.h, 4, max_points=64)
    G.makeK(**syn.KITTI00)
    with pytest.raises(RuntimeError):                  # no new frame yet
        G.traceSetPoints(P.u, P.v, P.energyTH, P.gradH, P.color, P.weights, P.host_idx)
        G.tracePoints(P.KRKi, P.Kt, P.aff, P.idepth_min, P.idepth_max, P.quality, P.status)
    G.set_new_image(P.image, 1.0)
    so = trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status)
    sg = G.tracePoints(P.KRKi, P.Kt, P.aff, P.idepth_min, P.idepth_max, P.quality, P.status)
    _same(sg, so)
    bad = P.host_idx.copy()
    bad[0] = 16
    with pytest.raises(RuntimeError):
        G.traceSetPoints(P.u, P.v, P.energyTH, P.gradH, P.color, P.weights, bad)
