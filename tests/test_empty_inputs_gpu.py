"""Empty inputs through every batched entry point of the C ABI: nothing to do must mean success (or the reference's own
degenerate behaviour), not a launch with a zero-sized grid or a crash."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)
F0 = np.zeros(0, np.float32)
I0 = np.zeros(0, np.int32)


def test_empty_batches(orc):
    from sdv_loam_amd import api, backend_api, reproject_api, synthetic as syn
    W = syn.make_window(w=320, h=200, nF=3, pts_per_kf=100, seed=31, calib=CAL)
    # tracker-side entry points
    T = api.CoarseTracker(W.w, W.h, 3, max_points=W.w * W.h, max_batch=2)
    O = orc.OracleTracker(W.w, W.h, 3)
    for X in (T, O):
        X.makeK(**CAL)
        X.set_new_image(W.images[0], 1.0)
        X.makeCoarseDepth(I0, I0, F0, F0)                      # no tuples -> empty template on every level
    assert all(len(T.get_ref(l)["u"]) == 0 == len(O.get_ref(l)["u"]) for l in range(3))
    T.traceSetPoints(F0, F0, F0, np.zeros((0, 4), np.float32), np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32), I0)
    st = T.tracePoints(np.zeros((1, 9), np.float32), np.zeros((1, 3), np.float32), np.zeros((1, 2), np.float32), F0, F0, F0, I0)
    assert len(st["status"]) == 0
    # structPoseEstimation without matches: num == 0 -> resOld = 0/0, nothing is ever accepted, pose unchanged (like the oracle)
    pose = np.array([0, 0, 0, 1, 0.1, 0.2, 0.3])
    hp = np.array([[0, 0, 0, 1, 0, 0, 0.0]])
    pg, tg, fg = T.structPoseEstimation(pose, F0, F0, F0, I0, hp, np.zeros((0, 2)))
    po, to, fo = O.structPoseEstimation(pose, F0, F0, F0, I0, hp, np.zeros((0, 2)))
    assert np.array_equal(pg, pose) and np.array_equal(po, pose) and len(tg) == len(to) and np.isnan(fg) and np.isnan(fo)
    # reprojector
    R = reproject_api.Reprojector(W.w, W.h, 3, max_frames=4, max_points=64)
    R.set_calib(**CAL)
    R.set_frame(0, hp[0], W.pyr0[0])
    R.set_cur(hp[0], syn.pyramid_numpy(W.images[1], 3))
    g = R.match(F0, F0, F0, I0, I0, I0)
    assert all(len(v) == 0 for v in g.values())
    # back end
    E = backend_api.EnergyFunctional(W.w, W.h, max_points=W.nP).load(W)
    r = E.optimizeImmature(I0, F0, F0, F0, F0, F0, np.zeros((0, 8), np.float32), np.zeros((0, 8), np.float32), np.zeros(0, np.uint8))
    assert len(r[0]) == 0 and r[2].shape == (0, W.nF)
