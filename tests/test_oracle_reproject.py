"""Known-answer tests pinning the CPU oracle's restatement of the Reprojector's per-candidate work (SURVEY.md 8f-2;
src/FullSystem/Reprojector.cpp).  The reference ships no tests for it; pins: an independent float64 numpy projection model, the
behaviour of the alignment on consistent synthetic data, and the structural properties of align1D / getBestSearchLevel.  CPU only."""
import numpy as np
import pytest

CAL = dict(fx=250., fy=252., cx=159.5, cy=99.5)


@pytest.fixture(scope="module")
def window():
    from sdv_loam_amd import synthetic as syn
    return syn.make_window(w=320, h=200, nF=4, pts_per_kf=250, seed=2, calib=CAL)


def _setup(P, cur_pose=None, cur_pyr=None, cur_exposure=1.0, cur_aff=(0.0, 0.0)):
    from oracle.reproject import OracleReprojector
    O = OracleReprojector(P.w, P.h, P.levels)
    O.set_calib(**P.calib)
    for k in range(len(P.frame_poses7)):
        O.set_frame(k, P.frame_poses7[k], P.frame_images[k])
    O.set_cur(P.cur_pose7 if cur_pose is None else cur_pose, P.cur_pyr if cur_pyr is None else cur_pyr, cur_exposure, *cur_aff)
    return O


def _project_np(P, pose_cur):
    from sdv_loam_amd import synthetic as syn
    fx, fy, cx, cy = (np.float64(np.float32(P.calib[k])) for k in ("fx", "fy", "cx", "cy"))
    w2c = syn._se3_inv_np(pose_cur)
    R, t = syn.quat_to_R(w2c[:4]), w2c[4:]
    out = np.zeros((P.n, 2))
    for i in range(P.n):
        hp = P.frame_poses7[P.host_idx[i]]
        X = syn.quat_to_R(hp[:4]) @ (np.array([(np.float64(P.u[i]) - cx) / fx, (np.float64(P.v[i]) - cy) / fy, 1.0])
                                     * np.float64(np.float32(1) / P.idepth[i])) + hp[4:]
        Y = R @ X + t
        out[i] = [fx * Y[0] / Y[2] + cx, fy * Y[1] / Y[2] + cy]
    return out


def test_project_matches_numpy_model(window):
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(window, levels=3)
    O = _setup(P)
    px, cell, q = O.project(P.u, P.v, P.idepth, P.host_idx)
    ref = _project_np(P, P.cur_pose7)
    assert np.abs(px - ref).max() < 1e-9
    inside = (px[:, 0].astype(int) >= 8) & (px[:, 0].astype(int) < P.w - 8) & (px[:, 1].astype(int) >= 8) & (px[:, 1].astype(int) < P.h - 8)
    assert np.array_equal(cell >= 0, inside) and 0 < inside.sum() < P.n
    n_cols = -(-P.w // 25)
    assert np.array_equal(cell[inside], (px[inside, 1] / 25).astype(int) * n_cols + (px[inside, 0] / 25).astype(int))
    for i in range(0, P.n, 37):                          # pointQualityComparator's key
        d = P.frame_images[P.host_idx[i]].reshape(-1, 3)[int(np.float32(P.v[i] * np.float32(P.w) + P.u[i]))]
        assert q[i] == np.sqrt(np.float32(d[1] * d[1] + d[2] * d[2]))


def test_alignment_recovers_the_true_position(window):
    """Current pose off by ~1 cm / 0.06 deg: the aligned positions are closer to the ground-truth projections than the
    predicted ones, and with the exact pose the alignment stays put."""
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(window, levels=3, pose_err=(0.02, 0.002))
    O = _setup(P)
    px0, cell, _ = O.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, lvl = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    gt = _project_np(P, P.gt_cur_pose7)
    use = ok & (cell >= 0) & (P.type == 0)
    assert use.sum() > 0.4 * P.n
    e0 = np.linalg.norm(px0[use] - gt[use], axis=1)
    e1 = np.linalg.norm(pm[use] - gt[use], axis=1)
    assert np.median(e0) > 0.3 and np.median(e1) < 0.35 * np.median(e0)
    O2 = _setup(P, cur_pose=P.gt_cur_pose7)
    px0g, cellg, _ = O2.project(P.u, P.v, P.idepth, P.host_idx)
    okg, pmg, _ = O2.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0g)
    useg = okg & (cellg >= 0)
    assert np.median(np.linalg.norm(pmg[useg] - px0g[useg], axis=1)) < 0.12
    assert np.all(lvl[cell >= 0][ok[cell >= 0]] == 0)          # no scale change between the key-frames and the new frame


def test_edgelets_move_along_the_warped_gradient_only(window):
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(window, levels=3, pose_err=(0.02, 0.002), edgelet_frac=1.0)
    O = _setup(P)
    px0, cell, _ = O.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, _ = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    okc, pmc, _ = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, np.zeros(P.n, np.int32), px0)
    use = ok & okc & (cell >= 0)
    assert use.sum() > 50
    d1 = pm[use] - px0[use]
    d2 = pmc[use] - px0[use]
    # 1-D search: the displacement is the corner displacement projected on (roughly) one direction -> never longer, usually shorter
    assert np.median(np.linalg.norm(d1, axis=1)) <= np.median(np.linalg.norm(d2, axis=1)) + 1e-6
    # and it is a pure line search: repeating it from the result moves (almost) nothing
    ok2, pm2, _ = O.find_match(P.u[use], P.v[use], P.idepth[use], P.host_idx[use], P.ref_idx[use], P.type[use], pm[use])
    assert np.median(np.linalg.norm(pm2[ok2] - pm[use][ok2], axis=1)) < 0.05


def test_search_level_follows_the_area_change(window):
    """getBestSearchLevel (:38-51): level = number of times det(A_cur_ref) can be quartered while > 3.  Shrinking the key-frame
    focal length by s makes the new frame see the patch s^2 times larger in area."""
    from sdv_loam_amd import synthetic as syn
    from oracle.reproject import OracleReprojector
    P = syn.make_reproject_problem(window, levels=3, pose_err=(0.0, 0.0))
    cur_hi = syn.pyramid_numpy(np.kron(window.images[-1], np.ones((1, 1), np.float32)), 3)
    for s, want in ((1.0, 0), (2.0, 1), (4.0, 2), (8.0, 2)):
        # put the new camera s times closer to every point by scaling the scene instead: idepth * s with poses' translations / s
        O = OracleReprojector(P.w, P.h, 3)
        O.set_calib(**P.calib)
        for k in range(len(P.frame_poses7)):
            pose = P.frame_poses7[k].copy()
            O.set_frame(k, pose, P.frame_images[k])
        # move the new camera towards the plane along its optical axis: area change ~ (d / (d - dz))^2
        cur = P.gt_cur_pose7.copy()
        R = syn.quat_to_R(cur[:4])
        depth = 1.0 / np.median(P.idepth)
        cur[4:] = cur[4:] + R @ np.array([0, 0, depth * (1 - 1 / s)])
        O.set_cur(cur, cur_hi)
        px0, cell, _ = O.project(P.u, P.v, P.idepth, P.host_idx)
        ok, pm, lvl = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
        sel = (cell >= 0) & (lvl >= 0)
        if s == 1.0:
            assert np.all(lvl[sel] == 0)
        else:
            assert sel.sum() > 0 and np.bincount(lvl[sel]).argmax() == want


def test_brightness_transfer_is_compensated(window):
    """The new frame is a*I + b of the key-frames' brightness; with the matching exposure / aff_g2l (AffLight::fromToVecExposure,
    :253-255) the alignment gives (nearly) the same positions as on the unmodified image."""
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(window, levels=3, pose_err=(0.01, 0.001))
    O = _setup(P)
    px0, cell, _ = O.project(P.u, P.v, P.idepth, P.host_idx)
    ok, pm, _ = O.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    a, b = 0.8, 12.0
    pyr = syn.pyramid_numpy((a * window.images[-1] + b).astype(np.float32), 3)
    O2 = _setup(P, cur_pyr=pyr, cur_exposure=1.0, cur_aff=(np.log(a), b))     # a_cur - a_ref = log(a), b_cur = b
    ok2, pm2, _ = O2.find_match(P.u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    both = ok & ok2 & (cell >= 0)
    assert both.sum() > 0.8 * (ok & (cell >= 0)).sum()
    assert np.median(np.linalg.norm(pm2[both] - pm[both], axis=1)) < 0.05


def test_border_and_degenerate_candidates_fail(window):
    from sdv_loam_amd import synthetic as syn
    P = syn.make_reproject_problem(window, levels=3)
    O = _setup(P)
    u = P.u.copy()
    u[:10] = 3.0                                      # reference pixel inside the 6-px border of isInFrame(px, halfpatch+2)
    px0, cell, _ = O.project(u, P.v, P.idepth, P.host_idx)
    ok, pm, lvl = O.find_match(u, P.v, P.idepth, P.host_idx, P.ref_idx, P.type, px0)
    assert not ok[:10].any() and np.all(lvl[:10] == -1) and np.array_equal(pm[:10], px0[:10])
