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
