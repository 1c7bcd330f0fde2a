"""Synthetic inputs for the hot path: band-limited noise images, reference point
sets per pyramid level, ground-truth motions.  Pure numpy/scipy -- used by tests/ and bench.py to feed the
SAME arrays to the HIP path and to the CPU oracle.  Nothing here is on the product's compute path.
"""
import numpy as np

KITTI00 = dict(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157)          # calib/KITTI/00.txt:1
KITTI360 = dict(fx=552.554261, fy=552.554261, cx=682.049453, cy=238.769549)  # calib/kitti_360.txt:1


def make_image(w, h, seed=0):
    """Band-limited noise: 3 octaves of Gaussian-blurred uniform noise (sigma 8, 3, 1 px), rescaled to [20,235]."""
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w), np.float64)
    for sigma, amp in ((8.0, 1.0), (3.0, 0.5), (1.0, 0.25)):
        o = gaussian_filter(rng.random((h, w)), sigma, mode="reflect")
        o = (o - o.mean()) / (o.std() + 1e-12)
        img += amp * o
    img = (img - img.min()) / (img.max() - img.min())
    return (20.0 + 215.0 * img).astype(np.float32)


def pyramid_numpy(color, levels):
    """numpy mirror of FrameHessian::makeImages (HessianBlocks.cpp:107-167): list of (h_l, w_l, 3) float32
    AoS {I,dx,dy}; gradient rows 0 and h_l-1 are left at 0 here (uninitialised in the reference)."""
    I = np.ascontiguousarray(color, np.float32)
    out = []
    for lvl in range(levels):
        if lvl > 0:
            P = out[-1][..., 0]
            hl, wl = P.shape[0] // 2, P.shape[1] // 2
            a = P[0:2 * hl:2, 0:2 * wl:2]
            b = P[0:2 * hl:2, 1:2 * wl:2]
            c = P[1:2 * hl:2, 0:2 * wl:2]
            d = P[1:2 * hl:2, 1:2 * wl:2]
            I = (np.float32(0.25) * (((a + b) + c) + d)).astype(np.float32)
        hl, wl = I.shape
        flat = I.reshape(-1)
        dx = np.zeros(hl * wl, np.float32)
        dy = np.zeros(hl * wl, np.float32)
        idx = np.arange(wl, wl * (hl - 1))
        dx[idx] = np.float32(0.5) * (flat[idx + 1] - flat[idx - 1])
        dy[idx] = np.float32(0.5) * (flat[idx + wl] - flat[idx - wl])
        dx[~np.isfinite(dx)] = 0
        dy[~np.isfinite(dy)] = 0
        out.append(np.stack([I, dx.reshape(hl, wl), dy.reshape(hl, wl)], axis=-1).astype(np.float32))
    return out


def level_intrinsics(calib, levels):
    """numpy mirror of CoarseTracker::makeK (CoarseTracker.cpp:77-106), float32 results."""
    fx = [np.float32(calib["fx"])]
    fy = [np.float32(calib["fy"])]
    cx = [np.float32(calib["cx"])]
    cy = [np.float32(calib["cy"])]
    for l in range(1, levels):
        fx.append(np.float32(np.float64(fx[l - 1]) * 0.5))
        fy.append(np.float32(np.float64(fy[l - 1]) * 0.5))
        cx.append(np.float32((np.float64(cx[0]) + 0.5) / (1 << l) - 0.5))
        cy.append(np.float32((np.float64(cy[0]) + 0.5) / (1 << l) - 0.5))
    return np.array(fx), np.array(fy), np.array(cx), np.array(cy)


def _bilinear64(img, x, y):
    ix = np.floor(x).astype(np.int64)
    iy = np.floor(y).astype(np.int64)
    dx = x - ix
    dy = y - iy
    return ((1 - dx) * (1 - dy) * img[iy, ix] + dx * (1 - dy) * img[iy, ix + 1]
            + (1 - dx) * dy * img[iy + 1, ix] + dx * dy * img[iy + 1, ix + 1])


def quat_to_R(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def se3_exp_np(xi):
    """Independent numpy SE(3) exponential, tangent [upsilon, omega] -> pose7 [qx qy qz qw tx ty tz]."""
    ups = np.asarray(xi[:3], np.float64)
    om = np.asarray(xi[3:], np.float64)
    th = np.linalg.norm(om)
    O = np.array([[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]])
    if th < 1e-10:
        V = np.eye(3) + 0.5 * O
        q = np.concatenate([0.5 * om, [1.0]])
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * O + (th - np.sin(th)) / th ** 3 * (O @ O)
        q = np.concatenate([np.sin(th / 2) / th * om, [np.cos(th / 2)]])
    return np.concatenate([q / np.linalg.norm(q), V @ ups])


class TrackerProblem:
    """One coarse-tracking problem: target pyramid source image + per-level reference point sets."""
    pass


def make_tracker_problem(w=1241, h=376, levels=4, n_points=2000, seed=0, calib=KITTI00,
                         gt_xi=None, gt_aff=(0.0, 0.0), ref_aff=(0.0, 0.0), image=None):
    """SURVEY.md 8d tracker inputs.

    The target image is band-limited noise B.  Reference points are random integer pixels of the level grid
    with idepth ~ U(0.02, 0.5); their reference colour is chosen so that the photometric residual is exactly
    zero at the ground-truth motion ``gt_xi`` (se3 tangent, ref->new) and affine ``gt_aff`` (a, b of the new
    frame):  c_ref = (B(pi(T_gt x)) - b_rel) / a_rel.  With gt_xi=None the ground truth is identity.
    """
    rng = np.random.default_rng(seed + 1000)
    P = TrackerProblem()
    P.w, P.h, P.levels, P.calib = w, h, levels, dict(calib)
    P.image = make_image(w, h, seed) if image is None else np.ascontiguousarray(image, np.float32)
    P.pyr = pyramid_numpy(P.image, levels)
    fx, fy, cx, cy = level_intrinsics(calib, levels)
    P.fx, P.fy, P.cx, P.cy = fx, fy, cx, cy
    P.gt_pose = se3_exp_np(np.zeros(6) if gt_xi is None else gt_xi)
    P.gt_aff = tuple(gt_aff)
    P.ref_aff = tuple(ref_aff)
    R = quat_to_R(P.gt_pose[:4])
    t = P.gt_pose[4:]
    a_rel = np.exp(gt_aff[0] - ref_aff[0])
    b_rel = gt_aff[1] - a_rel * ref_aff[1]
    P.ref = []
    for l in range(levels):
        wl, hl = w >> l, h >> l
        I = P.pyr[l][..., 0].astype(np.float64)
        g2 = P.pyr[l][..., 1].astype(np.float64) ** 2 + P.pyr[l][..., 2].astype(np.float64) ** 2
        # oversample candidates, prefer |grad|^2 > 50, keep only those whose GT projection is inside
        m = max(4 * n_points, 64)
        xs = rng.integers(4, wl - 4, size=m).astype(np.float64)
        ys = rng.integers(4, hl - 4, size=m).astype(np.float64)
        idp = rng.uniform(0.02, 0.5, size=m)
        Ki = np.array([[1 / fx[l], 0, -cx[l] / fx[l]], [0, 1 / fy[l], -cy[l] / fy[l]], [0, 0, 1]], np.float64)
        pt = (R @ Ki @ np.stack([xs, ys, np.ones(m)])) + t[:, None] * idp[None, :]
        Ku = fx[l] * pt[0] / pt[2] + cx[l]
        Kv = fy[l] * pt[1] / pt[2] + cy[l]
        inside = (Ku > 3) & (Kv > 3) & (Ku < wl - 4) & (Kv < hl - 4) & (pt[2] > 0)
        strong = g2[ys.astype(int), xs.astype(int)] > 50
        order = np.argsort(~(inside & strong), kind="stable")  # strong&inside first
        order = order[inside[order]][:n_points]
        order = np.sort(order)
        xs, ys, idp, Ku, Kv = xs[order], ys[order], idp[order], Ku[order], Kv[order]
        col = (_bilinear64(I, Ku, Kv) - b_rel) / a_rel
        P.ref.append(dict(u=xs.astype(np.float32), v=ys.astype(np.float32),
                          idepth=idp.astype(np.float32), color=col.astype(np.float32)))
    return P


def perturbation(seed=0, sigma_t=0.05, sigma_r=0.005):
    rng = np.random.default_rng(seed + 2000)
    return np.concatenate([rng.normal(0, sigma_t, 3), rng.normal(0, sigma_r, 3)])


# =====================================================================================================
# Sliding-window back end: nF key-frames looking at a slanted textured plane.
# =====================================================================================================
PATTERN8 = np.array([[0, -2], [-1, -1], [1, -1], [-2, 0], [0, 0], [2, 0], [-1, 1], [0, 2]], np.int32)  # settings.cpp:250


class Window:
    pass


def _se3_inv_np(p):
    q = np.array([-p[0], -p[1], -p[2], p[3]])
    R = quat_to_R(q)
    return np.concatenate([q, -R @ p[4:]])


def _se3_mul_np(a, b):
    ax, ay, az, aw = a[:4]
    bx, by, bz, bw = b[:4]
    q = np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                  aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])
    q /= np.linalg.norm(q)
    return np.concatenate([q, a[4:] + quat_to_R(a[:4]) @ b[4:]])


def make_window(w=1241, h=376, nF=8, pts_per_kf=2000, seed=0, calib=KITTI00, sensor_frac=0.3, matcher_sigma=0.3,
                state_sigma=2e-4, idepth_sigma=0.003, evalpt_sigma=(2e-3, 2e-4), spacing=1.0, marg_prior=True):
    """Synthetic EnergyFunctional window.  Ground truth: camera k at z = k*spacing with a small yaw, all frames see the
    plane n.X = d; frame images are rendered from one texture so that the 8-pixel photometric outlier test of
    linearize() sees consistent data.  Returns a Window with flat numpy arrays in the reference's iteration order
    (frames -> points hosted in the frame -> one residual per other frame)."""
    from scipy.ndimage import gaussian_filter, map_coordinates
    rng = np.random.default_rng(seed + 5000)
    W = Window()
    W.w, W.h, W.nF, W.calib = w, h, nF, dict(calib)
    fx, fy, cx, cy = (np.float64(np.float32(calib[k])) for k in ("fx", "fy", "cx", "cy"))
    # texture on the plane, 40 px/m, +-25 m
    tex_res, tex_half = 40.0, 25.0
    tn = int(2 * tex_half * tex_res)
    tex = np.zeros((tn, tn))
    for sigma, amp in ((12.0, 1.0), (4.0, 0.6), (1.8, 0.35)):
        o = gaussian_filter(rng.random((tn, tn)), sigma, mode="wrap")
        tex += amp * (o - o.mean()) / o.std()
    tex = 20.0 + 215.0 * (tex - tex.min()) / (tex.max() - tex.min())
    n = np.array([0.25, 0.45, 1.0])
    n /= np.linalg.norm(n)
    d = 14.0
    e1 = np.cross(n, [0, 1, 0]); e1 /= np.linalg.norm(e1)
    e2 = np.cross(n, e1)

    # ground-truth world->cam poses
    gt = []
    for k in range(nF):
        yaw = 0.01 * k
        camToWorld = se3_exp_np([0.05 * np.sin(k), 0.02 * k, spacing * k, 0.0, yaw, 0.002 * k])
        gt.append(_se3_inv_np(camToWorld))
    W.gt_worldToCam = np.array(gt)

    def render(p):  # image of frame with world->cam pose p, and its depth map
        R = quat_to_R(p[:4]); t = p[4:]
        C = -R.T @ t
        ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
        rays = np.stack([(xs - cx) / fx, (ys - cy) / fy, np.ones_like(xs)], -1) @ R  # rows: R^T r
        lam = (d - n @ C) / (rays @ n)
        X = C + lam[..., None] * rays
        s = (X @ e1 + tex_half) * tex_res
        tt = (X @ e2 + tex_half) * tex_res
        img = map_coordinates(tex, [tt, s], order=1, mode="nearest")
        return img.astype(np.float32), lam  # lam == depth z in the camera (ray has z=1)

    W.images, depths = [], []
    for k in range(nF):
        img, z = render(gt[k])
        W.images.append(img)
        depths.append(z)
    W.pyr0 = [pyramid_numpy(img, 1)[0] for img in W.images]  # level-0 AoS {I,dx,dy}

    # frames: evaluation point = GT perturbed; state = small non-zero increment on top; state_zero = 0 except b
    W.evalPT = np.array([_se3_mul_np(se3_exp_np(np.concatenate([rng.normal(0, evalpt_sigma[0], 3), rng.normal(0, evalpt_sigma[1], 3)])), gt[k])
                         for k in range(nF)])
    W.evalPT[0] = gt[0]
    W.state = np.zeros((nF, 10))
    W.state[:, :6] = rng.normal(0, state_sigma, (nF, 6))
    W.state[0, :6] = 0
    W.state[:, 6] = 0.001 * rng.normal(0, 1, nF)   # a (scaled x10)
    W.state[:, 7] = 0.0005 * rng.normal(0, 1, nF)  # b (scaled x1000)
    W.state_zero = np.zeros((nF, 10))
    W.state_zero[:, 6:8] = W.state[:, 6:8]
    W.frameID = np.arange(nF, dtype=np.int32)
    W.ab_exposure = np.ones(nF, np.float32)
    W.frameEnergyTH = np.full(nF, 8 * 8 * 8, np.float32)  # FrameHessian ctor: 8*8*patternNum
    W.value_scaled = np.array([fx, fy, cx, cy], np.float64)
    W.value_minus_value_zero = np.array([1e-3, -5e-4, 2e-3, 1e-3]) / np.array([50., 50., 50., 50.])

    # points + residuals
    host, us, vs, idp, col, wts, prior, sensor = [], [], [], [], [], [], [], []
    r_point, r_target, r_match = [], [], []
    for hk in range(nF):
        I = W.pyr0[hk]
        g2 = I[..., 1].astype(np.float64) ** 2 + I[..., 2].astype(np.float64) ** 2
        m = 6 * pts_per_kf
        xs = rng.integers(8, w - 8, m)
        ys = rng.integers(8, h - 8, m)
        order = np.argsort(~(g2[ys, xs] > 30), kind="stable")[:pts_per_kf]
        order = np.sort(order)
        xs, ys = xs[order], ys[order]
        z = depths[hk][ys, xs]
        true_id = 1.0 / z
        Rh = quat_to_R(gt[hk][:4]); th = gt[hk][4:]
        Xc = np.stack([(xs - cx) / fx * z, (ys - cy) / fy * z, z], -1)
        Xw = (Xc - th) @ Rh  # R^T (Xc - t)
        base = len(host)
        for i in range(len(xs)):
            host.append(hk)
        us += list(xs.astype(np.float32)); vs += list(ys.astype(np.float32))
        idp += list((true_id * (1 + rng.normal(0, idepth_sigma, len(xs)))).astype(np.float32))
        px = xs[:, None] + PATTERN8[None, :, 0]
        py = ys[:, None] + PATTERN8[None, :, 1]
        col += list(I[py, px, 0].astype(np.float32))
        # ImmaturePoint's constructor (ImmaturePoint.cpp:14-27), float arithmetic: the pattern pixels go through
        # getInterpolatedElement33BiLin (globalFuncs.h:142-166), whose "gradient" at an integer position is the FORWARD difference
        # (I[x+1] - I[x], I[y+1] - I[y]) -- not the stored central differences; weights = sqrtf(c / (c + gx*gx + gy*gy)), c = 50*50
        I0 = I[..., 0]
        gx = (I0[py, px + 1] - I0[py, px]).astype(np.float32)
        gy = (I0[py + 1, px] - I0[py, px]).astype(np.float32)
        g2f = (gx * gx + gy * gy).astype(np.float32)
        wts += list(np.sqrt(np.float32(2500.0) / (np.float32(2500.0) + g2f)).astype(np.float32))
        prior += [hk == 0] * len(xs)
        sensor += list(rng.random(len(xs)) < sensor_frac)
        for tk in range(nF):
            if tk == hk:
                continue
            Rt = quat_to_R(gt[tk][:4]); tt = gt[tk][4:]
            Xt = Xw @ Rt.T + tt
            proj = np.stack([fx * Xt[:, 0] / Xt[:, 2] + cx, fy * Xt[:, 1] / Xt[:, 2] + cy], -1)
            r_match.append((tk, proj + rng.normal(0, matcher_sigma, proj.shape)))
        # residual order: point-major, targets ascending
        for i in range(len(xs)):
            for j, tk in enumerate([t for t in range(nF) if t != hk]):
                r_point.append(base + i)
                r_target.append(tk)
        W._match_blocks = getattr(W, "_match_blocks", []) + [np.stack([mm[1] for mm in r_match[-(nF - 1):]], 1)]  # (P, nF-1, 2)
    W.host = np.array(host, np.int32)
    W.u = np.array(us, np.float32); W.v = np.array(vs, np.float32)
    W.idepth = np.array(idp, np.float32)
    W.idepth_zero = W.idepth.copy()
    W.color = np.array(col, np.float32).reshape(-1, 8)
    W.weights = np.array(wts, np.float32).reshape(-1, 8)
    W.hasDepthPrior = np.array(prior, np.uint8)
    W.isFromSensor = np.array(sensor, np.uint8)
    W.r_point = np.array(r_point, np.int32)
    W.r_target = np.array(r_target, np.int32)
    W.r_matcher = np.concatenate([b.reshape(-1, 2) for b in W._match_blocks]).astype(np.float64)
    del W._match_blocks
    nR = len(W.r_point)
    W.r_state = np.zeros(nR, np.int32)          # ResState::IN after resetOOB()
    W.r_hasMatcher = np.ones(nR, np.uint8)
    W.r_isLinearized = np.zeros(nR, np.uint8)
    W.r_isActive = np.zeros(nR, np.uint8)
    W.nP, W.nR = len(W.host), nR
    ndim = 4 + 6 * nF
    if marg_prior:
        A = rng.normal(0, 1, (ndim, ndim))
        W.HM = 1e2 * (A @ A.T) / ndim
        W.bM = rng.normal(0, 10, ndim)
    else:
        W.HM = np.zeros((ndim, ndim)); W.bM = np.zeros(ndim)
    return W


def subwindow(W, frames, points, residual_mask=None, HM=None, bM=None):
    """The window restricted to `frames` (indices into W, in the order given) and `points` (indices into W, in the order given -- must be
    grouped by host frame in the order of `frames`): residuals whose point and target both survive (and residual_mask allows) follow, in
    point-major order.  What an EnergyFunctional looks like after removePoint / dropResidual / marginalizeFrame -- or, read the other way,
    before insertPoint / insertResidual / insertFrame: both ends of a key-frame update are sub-windows of one larger synthetic window."""
    import copy
    frames = np.asarray(frames, int)
    points = np.asarray(points, int)
    S = copy.copy(W)
    fmap = -np.ones(W.nF, int); fmap[frames] = np.arange(len(frames))
    pmap = -np.ones(W.nP, int); pmap[points] = np.arange(len(points))
    assert (fmap[W.host[points]] >= 0).all() and (np.diff(fmap[W.host[points]]) >= 0).all(), "points must be grouped by host in frame order"
    S.nF = len(frames)
    for name in ("evalPT", "state", "state_zero", "frameID", "ab_exposure", "frameEnergyTH", "gt_worldToCam"):
        setattr(S, name, getattr(W, name)[frames].copy())
    S.images = [W.images[k] for k in frames]
    S.pyr0 = [W.pyr0[k] for k in frames]
    for name in ("u", "v", "idepth", "idepth_zero", "color", "weights", "hasDepthPrior", "isFromSensor"):
        setattr(S, name, getattr(W, name)[points].copy())
    S.host = fmap[W.host[points]].astype(np.int32)
    keep = (pmap[W.r_point] >= 0) & (fmap[W.r_target] >= 0)
    if residual_mask is not None:
        keep &= np.asarray(residual_mask, bool)
    ridx = np.nonzero(keep)[0]
    ridx = ridx[np.argsort(pmap[W.r_point[ridx]], kind="stable")]      # point-major in the NEW point order
    S.r_src = ridx                                                      # index of every residual in W
    S.r_point = pmap[W.r_point[ridx]].astype(np.int32)
    S.r_target = fmap[W.r_target[ridx]].astype(np.int32)
    for name in ("r_matcher", "r_state", "r_hasMatcher", "r_isLinearized", "r_isActive"):
        setattr(S, name, getattr(W, name)[ridx].copy())
    S.nP, S.nR = len(points), len(ridx)
    n = 4 + 6 * S.nF
    S.HM = np.zeros((n, n)) if HM is None else np.asarray(HM, np.float64)
    S.bM = np.zeros(n) if bM is None else np.asarray(bM, np.float64)
    S.p_src, S.f_src = points, frames
    return S


# =====================================================================================================
# structPoseEstimation: map points hosted in the window's key-frames, matched 2-D positions
# in the current frame, and a perturbed initial camToWorld of the current frame.
# =====================================================================================================
class StructProblem:
    pass


def make_struct_problem(n=1200, n_hosts=7, w=1241, h=376, seed=0, calib=KITTI00, noise_px=0.3, outlier_frac=0.05,
                        pose_err=(0.05, 0.004), oob_frac=0.02):
    """Returns StructProblem with u,v,idepth (float32, host pixel + inverse depth), host_idx (int32), host_poses7
    (camToWorld per host, Sophus data() layout), obs (n x 2 float64 matched pixels in the current frame),
    gt_curToWorld7 and init_curToWorld7."""
    rng = np.random.default_rng(seed + 9000)
    P = StructProblem()
    P.w, P.h, P.calib, P.n = w, h, dict(calib), n
    fx, fy, cx, cy = (np.float64(np.float32(calib[k])) for k in ("fx", "fy", "cx", "cy"))
    hosts = []
    for k in range(n_hosts):
        xi = np.array([0.05 * rng.normal(), 0.02 * rng.normal(), 0.9 * k, 0.004 * rng.normal(), 0.01 * rng.normal(), 0.004 * rng.normal()])
        hosts.append(se3_exp_np(xi))                       # camToWorld
    P.host_poses7 = np.array(hosts)
    gt = se3_exp_np(np.array([0.03, -0.02, 0.9 * n_hosts, 0.003, -0.012, 0.002]))
    P.gt_curToWorld7 = gt
    gt_w2c = _se3_inv_np(gt)
    Rc, tc = quat_to_R(gt_w2c[:4]), gt_w2c[4:]
    P.host_idx = rng.integers(0, n_hosts, n).astype(np.int32)
    u = rng.uniform(8, w - 8, n)
    v = rng.uniform(8, h - 8, n)
    depth = rng.uniform(6.0, 45.0, n) + 0.9 * (n_hosts - P.host_idx)
    P.u, P.v, P.idepth = u.astype(np.float32), v.astype(np.float32), (1.0 / depth).astype(np.float32)
    obs = np.zeros((n, 2))
    for i in range(n):
        hp = hosts[P.host_idx[i]]
        X = quat_to_R(hp[:4]) @ (np.array([(P.u[i] - cx) / fx, (P.v[i] - cy) / fy, 1.0]) / np.float64(P.idepth[i])) + hp[4:]
        Y = Rc @ X + tc
        obs[i] = [fx * Y[0] / Y[2] + cx, fy * Y[1] / Y[2] + cy]
    obs += rng.normal(0, noise_px, obs.shape)
    out = rng.random(n) < outlier_frac
    obs[out] += rng.normal(0, 25.0, (int(out.sum()), 2))
    # a few matches whose map point leaves the current image (Ku >= wM3G fails world2frame's bounds test): near the right
    # border of the host and close to the camera, so that the forward motion pushes it out of view
    oob = rng.random(n) < oob_frac
    P.u[oob] = np.float32(w - 9.0)
    P.idepth[oob] = np.float32(1.0 / 8.0)
    P.obs = obs
    P.init_curToWorld7 = _se3_mul_np(gt, se3_exp_np(np.concatenate([rng.normal(0, pose_err[0], 3), rng.normal(0, pose_err[1], 3)])))
    return P


# =====================================================================================================
# Reprojector: the last key-frame of a synthetic window plays the new frame; the active
# points of the other key-frames are the candidates.
# =====================================================================================================
class ReprojectProblem:
    pass


def make_reproject_problem(W, levels=3, seed=0, pose_err=(0.01, 0.001), edgelet_frac=0.3):
    """W: a Window from make_window().  Returns ReprojectProblem with frames (camToWorld7, level-0 AoS image) for key-frames
    0..nF-2, the current frame (camToWorld7 perturbed from the ground truth so that the alignment has something to do, image
    pyramid), and the candidate points (u, v, idepth, host_idx, type; ref_idx = host_idx as in a window of > 2 frames)."""
    rng = np.random.default_rng(seed + 12000)
    P = ReprojectProblem()
    P.w, P.h, P.levels, P.calib = W.w, W.h, levels, dict(W.calib)
    nK = W.nF - 1
    P.frame_poses7 = np.array([_se3_inv_np(W.gt_worldToCam[k]) for k in range(nK)])
    P.frame_images = [W.pyr0[k] for k in range(nK)]
    P.frame_exposure = np.ones(nK, np.float32)
    P.frame_aff = np.zeros((nK, 2))
    gt_cur = _se3_inv_np(W.gt_worldToCam[nK])
    P.gt_cur_pose7 = gt_cur
    P.cur_pose7 = _se3_mul_np(gt_cur, se3_exp_np(np.concatenate([rng.normal(0, pose_err[0], 3), rng.normal(0, pose_err[1], 3)])))
    P.cur_pyr = pyramid_numpy(W.images[nK], levels)
    P.cur_exposure, P.cur_aff = np.float32(1.0), (0.0, 0.0)
    sel = W.host < nK
    P.u, P.v, P.idepth = W.u[sel].copy(), W.v[sel].copy(), W.idepth[sel].copy()
    P.host_idx = W.host[sel].astype(np.int32)
    P.ref_idx = P.host_idx.copy()
    P.type = (rng.random(int(sel.sum())) < edgelet_frac).astype(np.int32)      # 0 = CORNER, 1 = EDGELET (HessianBlocks.h:401)
    P.n = int(sel.sum())
    return P


# =====================================================================================================
# ImmaturePoint::traceOn: the window's points re-used as immature points of key-frames
# 0..nF-2, traced on the last frame.
# =====================================================================================================
class TraceProblem:
    pass


IPS_GOOD, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED = range(6)   # ImmaturePoint.h:20-30


def make_trace_problem(W, target=None, pose_err=(0.0, 0.0), seed=0):
    """Immature points = the points of key-frames != target, with the fields the ImmaturePoint constructor computes
    (ImmaturePoint.cpp:8-35): color, weights, gradH, energyTH; initial state idepth_min = 0, idepth_max = NaN, quality = 10000,
    status UNINITIALIZED.  Per-host KRKi / Kt / aff as FullSystem::traceNewCoarse builds them (FullSystem.cpp:525-538)."""
    rng = np.random.default_rng(seed + 15000)
    P = TraceProblem()
    tgt = W.nF - 1 if target is None else target
    P.w, P.h, P.target = W.w, W.h, tgt
    fx, fy, cx, cy = (np.float32(W.calib[k]) for k in ("fx", "fy", "cx", "cy"))
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    Ki = np.linalg.inv(K.astype(np.float64)).astype(np.float32)
    sel = W.host != tgt
    P.n = int(sel.sum())
    P.u, P.v = W.u[sel].copy(), W.v[sel].copy()
    P.host_idx = W.host[sel].astype(np.int32)
    P.color, P.weights = W.color[sel].copy(), W.weights[sel].copy()
    P.true_idepth = W.idepth[sel].copy()
    gradH = np.zeros((P.n, 4), np.float32)
    for i in range(P.n):
        I = W.pyr0[P.host_idx[i]]
        g = np.zeros((2, 2), np.float32)
        for dx, dy in PATTERN8:      # gradH += ptc.tail<2>() * ptc.tail<2>().transpose() with the forward differences of 33BiLin (ImmaturePoint.cpp:22-25)
            y0, x0 = int(P.v[i]) + dy, int(P.u[i]) + dx
            gv = np.array([I[y0, x0 + 1, 0] - I[y0, x0, 0], I[y0 + 1, x0, 0] - I[y0, x0, 0]], np.float32)
            g = (g + np.outer(gv, gv).astype(np.float32)).astype(np.float32)
        gradH[i] = g.reshape(-1)
    P.gradH = gradH
    P.energyTH = np.full(P.n, 8 * 12 * 12, np.float32)       # patternNum * setting_outlierTH * overallEnergyTHWeight^2
    P.idepth_min = np.zeros(P.n, np.float32)
    P.idepth_max = np.full(P.n, np.nan, np.float32)
    P.quality = np.full(P.n, 10000, np.float32)
    P.status = np.full(P.n, IPS_UNINITIALIZED, np.int32)
    w2c_t = _se3_mul_np(se3_exp_np(np.concatenate([rng.normal(0, pose_err[0], 3), rng.normal(0, pose_err[1], 3)])), W.gt_worldToCam[tgt])
    P.KRKi = np.zeros((W.nF, 9), np.float32)
    P.Kt = np.zeros((W.nF, 3), np.float32)
    P.aff = np.tile(np.array([1.0, 0.0], np.float32), (W.nF, 1))
    for hk in range(W.nF):
        hostToNew = _se3_mul_np(w2c_t, _se3_inv_np(W.gt_worldToCam[hk]))
        R = quat_to_R(hostToNew[:4]).astype(np.float32)
        P.KRKi[hk] = ((K @ R).astype(np.float32) @ Ki).astype(np.float32).reshape(-1)
        P.Kt[hk] = (K @ hostToNew[4:].astype(np.float32)).astype(np.float32)
    P.image = W.images[tgt]
    P.dI = W.pyr0[tgt]
    return P
