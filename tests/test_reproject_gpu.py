"""GPU parity of sdvgn_reproj_match (Reprojector::reprojectPoint / findMatchDirect / align1D / align2D,
src/FullSystem/Reprojector.cpp) against the CPU oracle, through the C ABI.  One lane runs one candidate with the reference's loop
order, so success flags, search levels and sub-pixel positions are expected bit-identical; the fp64 geometry is compared to 1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)


def _window(seed=2, n=250):
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=320, h=200, nF=4, pts_per_kf=n, seed=seed, calib=CAL)


def _pair(P, cur_pose=None, aff=(0.0, 0.0), frame_aff=None):
    from oracle.reproject import OracleReprojector
    from sdv_loam_amd import reproject_api
    O = OracleReprojector(P.w, P.h, P.levels)
    G = reproject_api.Reprojector(P.w, P.h, P.levels, max_frames=8, max_points=4096)
    for T in (O, G):
        T.set_calib(**P.calib)
        for k in range(len(P.frame_poses7)):
            fa = (0.0, 0.0) if frame_aff is None else frame_aff[k]
            T.set_frame(k, P.frame_poses7[k], P.frame_images[k], 1.0, fa[0], fa[1])
        T.set_cur(P.cur_pose7 if cur_pose is None else cur_pose, P.cur_pyr, 1.0, aff[0], aff[1])
    return G, O


def _compare(P, G, O):
    g = G.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    px0, cell, q = O.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, lvl = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    assert np.abs(g["px0"] - px0).max() < 1e-9
    assert np.array_equal(g["cell"], cell)
    assert np.array_equal(g["quality"], q)
    cand = cell >= 0                                      # findMatchDirect is only ever called for grid candidates
    assert np.array_equal(g["success"][cand], ok[cand]) and not g["success"][~cand].any()
    assert np.array_equal(g["level"][cand], lvl[cand])
    good = cand & ok
    assert good.sum() > 0
    assert np.array_equal(g["px"][good], pm[good])        # float arithmetic in the reference's order: bit-identical
    return g, good


@pytest.mark.parametrize("seed,pose_err,edge", [(2, (0.01, 0.001), 0.3), (3, (0.03, 0.003), 0.0), (4, (0.0, 0.0), 1.0), (5, (0.05, 0.006), 0.5)])
def test_match_parity(orc, seed, pose_err, edge):
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(_window(seed), levels=3, seed=seed, pose_err=pose_err, edgelet_frac=edge)
    G, O = _pair(P)
    g, good = _compare(P, G, O)
    assert good.sum() > 0.3 * P.n


def test_match_parity_scale_change_and_brightness(orc):
    from sdv_loam_amd import synthetic as syn
    W = _window(6)
    P = syn.make_reproject_problem(W, levels=3, seed=6, pose_err=(0.01, 0.001))
    cur = P.gt_cur_pose7.copy()
    cur[4:] = cur[4:] + syn.quat_to_R(cur[:4]) @ np.array([0, 0, (1.0 / np.median(P.idepth)) * 0.55])   # ~2.2x closer: level 1
    P.cur_pyr = syn.pyramid_numpy((0.85 * W.images[-1] + 9.0).astype(np.float32), 3)
    G, O = _pair(P, cur_pose=cur, aff=(np.log(0.85), 9.0), frame_aff=[(0.01 * k, 0.5 * k) for k in range(3)])
    g, good = _compare(P, G, O)
    assert (g["level"][g["cell"] >= 0] >= 1).any()


def test_selection_replays_reprojectMap(orc):
    """Host replay of the grid walk on GPU results == the same walk on oracle results (overlap_pts identical)."""
    from sdv_loam_amd import reproject_api, synthetic as syn
    P = syn.make_reproject_problem(_window(7, 400), levels=3, seed=7, pose_err=(0.02, 0.002))
    G, O = _pair(P)
    g = G.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    px0, cell, q = O.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, _ = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    n_cells = (-(-P.w // 25)) * (-(-P.h // 25))
    rng = np.random.default_rng(0)
    cell_order = rng.permutation(n_cells)
    # key-frames closest to the new frame first (:125-131), points in host order
    dist = [np.linalg.norm(P.cur_pose7[4:] - P.frame_poses7[k][4:]) for k in range(len(P.frame_poses7))]
    order = [i for k in np.argsort(dist, kind="stable") for i in np.nonzero(P.host_idx == k)[0]]
    a = reproject_api.select_matches(g["cell"], g["quality"], g["success"], g["px"], order, cell_order, 60)
    b = reproject_api.select_matches(cell, q, ok & (cell >= 0), pm, order, cell_order, 60)
    assert [i for i, _ in a] == [i for i, _ in b] and len(a) == 61
    assert all(np.array_equal(pa, pb) for (_, pa), (_, pb) in zip(a, b))


def test_errors_and_borrowed_device_images(orc):
    import ctypes as C
    import torch
    from sdv_loam_amd import reproject_api, synthetic as syn
    P = syn.make_reproject_problem(_window(8), levels=3, seed=8)
    G, O = _pair(P)
    ref = G.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    # same data through device pointers owned by somebody else (here: torch tensors)
    G2 = reproject_api.Reprojector(P.w, P.h, P.levels, max_frames=8, max_points=4096)
    G2.set_calib(**P.calib)
    keep = []
    for k in range(len(P.frame_poses7)):
        t = torch.from_numpy(np.ascontiguousarray(P.frame_images[k], np.float32).reshape(-1)).cuda()
        keep.append(t)
        G2.set_frame(k, P.frame_poses7[k], None, dev_ptr=C.c_void_p(t.data_ptr()))
    pyr = [torch.from_numpy(np.ascontiguousarray(img, np.float32).reshape(-1)).cuda() for img in P.cur_pyr]
    torch.cuda.synchronize()
    G2.set_cur(P.cur_pose7, dev_ptrs=[C.c_void_p(t.data_ptr()) for t in pyr])
    got = G2.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
    for k in ("px0", "cell", "quality", "success", "px", "level"):
        assert np.array_equal(got[k], ref[k]), k
    bad = P.host_idx.copy()
    bad[0] = 7                                            # frame never registered
    with pytest.raises(RuntimeError):
        G.match(P.u, P.v, P.idepth, bad, P.ref_idx, P.type)
    G3 = reproject_api.Reprojector(P.w, P.h, P.levels, max_frames=8, max_points=64)
    G3.set_calib(**P.calib)
    with pytest.raises(RuntimeError):                     # no current frame yet / too many points
        G3.match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type)
