"""Known-answer tests pinning the CPU oracle's restatement of ImmaturePoint::traceOn (SURVEY.md 8f-4;
src/FullSystem/ImmaturePoint.cpp:47-353).  The reference ships no tests for it; pins: the search finds the true projection /
inverse depth of synthetic points, the interval logic across frames, and every early-return branch.  CPU only."""
import numpy as np
import pytest

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)


@pytest.fixture(scope="module")
def window():
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=320, h=200, nF=4, pts_per_kf=250, seed=2, calib=CAL)


def _true_uv(P, idepth):
    """projection of the points at inverse depth `idepth` with the problem's own KRKi / Kt (float64)"""
    KRKi = P.KRKi[P.host_idx].reshape(-1, 3, 3).astype(np.float64)
    Kt = P.Kt[P.host_idx].astype(np.float64)
    pr = np.einsum("nij,nj->ni", KRKi, np.stack([P.u, P.v, np.ones_like(P.u)], -1).astype(np.float64))
    p = pr + Kt * idepth[:, None]
    return p[:, :2] / p[:, 2:3]


def test_uninitialized_points_find_their_match(window):
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    P = syn.make_trace_problem(window)
    st = trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status)
    assert set(np.unique(st["status"])) <= {syn.IPS_GOOD, syn.IPS_OOB, syn.IPS_OUTLIER}
    good = st["status"] == syn.IPS_GOOD
    assert good.sum() > 0.6 * P.n
    uv_true = _true_uv(P, P.true_idepth.astype(np.float64))
    uv0 = _true_uv(P, np.zeros(P.n))
    reach = np.linalg.norm(uv_true - uv0, axis=1) < (P.w + P.h) * 0.027 - 1        # inside the searched segment
    sel = good & reach
    err = np.linalg.norm(st["lastTraceUV"][sel] - uv_true[sel], axis=1)
    assert np.median(err) < 0.3 and np.mean(err < 1.0) > 0.85
    assert np.mean((st["idepth_min"][sel] <= P.true_idepth[sel] * 1.02) & (st["idepth_max"][sel] >= P.true_idepth[sel] * 0.98)) > 0.8
    assert np.all(st["idepth_min"][good] <= st["idepth_max"][good]) and np.all(st["idepth_max"][good] >= 0)
    assert np.all(st["interval"][good] >= 0.8 - 1e-6) and np.all(st["interval"][good] <= 20)      # 2*errorInPixel, errorInPixel in [0.4,10]
    bad = st["status"] != syn.IPS_GOOD
    assert np.all(st["lastTraceUV"][bad] == -1) and np.all(st["interval"][bad] == 0)
    assert np.all(st["quality"][good] > 0)


def test_second_frame_narrows_the_interval(window):
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    P1 = syn.make_trace_problem(window, target=2)          # short baseline first ...
    s1 = trace_on(P1, P1.dI, P1.idepth_min, P1.idepth_max, P1.quality, P1.status)
    common = np.nonzero(window.host < 2)[0]                              # points hosted in frames 0,1: traced on 2, then on 3
    P2 = syn.make_trace_problem(window, target=3)          # ... then a longer one: the same depth interval spans more pixels
    a = np.searchsorted(np.nonzero(window.host != 2)[0], common)         # their rows in P1 ...
    b = np.searchsorted(np.nonzero(window.host != 3)[0], common)         # ... and in P2
    imin, imax, q, stt = P2.idepth_min.copy(), P2.idepth_max.copy(), P2.quality.copy(), P2.status.copy()
    imin[b], imax[b], q[b], stt[b] = s1["idepth_min"][a], s1["idepth_max"][a], s1["quality"][a], s1["status"][a]
    s2 = trace_on(P2, P2.dI, imin, imax, q, stt)
    was_good = s1["status"][a] == syn.IPS_GOOD
    now = s2["status"][b][was_good]
    assert set(np.unique(now)) <= {syn.IPS_GOOD, syn.IPS_SKIPPED, syn.IPS_BADCONDITION, syn.IPS_OUTLIER, syn.IPS_OOB}
    assert np.mean(np.isin(now, [syn.IPS_GOOD, syn.IPS_SKIPPED, syn.IPS_BADCONDITION])) > 0.8
    g2 = was_good & (s2["status"][b] == syn.IPS_GOOD)
    w1 = (s1["idepth_max"][a] - s1["idepth_min"][a])[g2]
    w2 = (s2["idepth_max"][b] - s2["idepth_min"][b])[g2]
    assert g2.sum() > 10 and np.median(w2 / w1) < 1.0
    # OOB points are left exactly as they were (:49)
    oob = s1["status"][a] == syn.IPS_OOB
    for k in ("idepth_min", "idepth_max", "quality"):
        assert np.array_equal(s2[k][b][oob], s1[k][a][oob], equal_nan=True)
    assert np.all(s2["status"][b][oob] == syn.IPS_OOB)


def test_skipped_and_outlier_branches(window):
    from oracle.trace import trace_on
    from sdv_loam_amd import synthetic as syn
    P = syn.make_trace_problem(window)
    # already certain: interval narrower than setting_trace_slackInterval = 1.5 px -> SKIPPED, lastTraceUV = midpoint
    imin = (P.true_idepth * 0.999).astype(np.float32)
    imax = (P.true_idepth * 1.001).astype(np.float32)
    st = trace_on(P, P.dI, imin, imax, P.quality, np.zeros(P.n, np.int32))
    sk = st["status"] == syn.IPS_SKIPPED
    assert sk.sum() > 0.7 * P.n
    mid = 0.5 * (_true_uv(P, imin.astype(np.float64)) + _true_uv(P, imax.astype(np.float64)))
    assert np.abs(st["lastTraceUV"][sk] - mid[sk]).max() < 1e-2 and np.all(st["interval"][sk] < 1.5)
    assert np.array_equal(st["idepth_min"][sk], imin[sk]) and np.array_equal(st["idepth_max"][sk], imax[sk])
    # energy threshold NaN (constructor saw a non-finite colour, ImmaturePoint.cpp:20) -> OUTLIER, second time -> OOB (:312-322)
    P.energyTH[:] = np.nan
    s1 = trace_on(P, P.dI, P.idepth_min, P.idepth_max, P.quality, P.status)
    searched = s1["status"] != syn.IPS_OOB
    assert np.all(s1["status"][searched] == syn.IPS_OUTLIER)
    s2 = trace_on(P, P.dI, s1["idepth_min"], s1["idepth_max"], s1["quality"], s1["status"])
    assert np.all(s2["status"] == syn.IPS_OOB)
