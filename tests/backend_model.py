"""Independent float64 numpy model of the back end's geometry, used to pin the oracle (and so the GPU):
projection, relative/absolute Jacobians by finite differences, and the full (un-marginalised) normal equations."""
import numpy as np
from scipy.linalg import expm

from sdv_loam_amd import synthetic as syn

SCALE_XI_TRANS, SCALE_XI_ROT, SCALE_F, SCALE_C = 0.5, 1.0, 50.0, 50.0


def hat6(x):
    M = np.zeros((4, 4))
    w = x[3:]
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = x[:3]
    return M


def T4(p7):
    M = np.eye(4)
    M[:3, :3] = syn.quat_to_R(p7[:4])
    M[:3, 3] = p7[4:]
    return M


def project(Kv, T, u, v, idepth):
    """Kv = (fx,fy,cx,cy); T = 4x4 host->target; returns (Ku, Kv)."""
    fx, fy, cx, cy = Kv
    x = np.array([(u - cx) / fx, (v - cy) / fy, 1.0])
    p = T[:3, :3] @ x + T[:3, 3] * idepth
    return np.array([fx * p[0] / p[2] + cx, fy * p[1] / p[2] + cy])


def fd(f, x0, k, h):
    xp = np.array(x0, float); xm = np.array(x0, float)
    xp[k] += h; xm[k] -= h
    return (f(xp) - f(xm)) / (2 * h)


def relative_jacobians_fd(Kv, T0, u, v, idepth):
    """d(Ku,Kv)/d[xi(6) | c(4, internal = value/SCALE) | idepth] by central differences, T = exp(xi) T0."""
    def f(p):
        Kp = np.array(Kv) + np.array([SCALE_F, SCALE_F, SCALE_C, SCALE_C]) * p[6:10]
        return project(Kp, expm(hat6(p[:6])) @ T0, u, v, idepth + p[10])
    steps = [1e-5] * 3 + [1e-6] * 3 + [1e-5] * 4 + [1e-6]
    return np.stack([fd(f, np.zeros(11), k, steps[k]) for k in range(11)], axis=1)  # 2 x 11


def abs_jacobian_rows(W, adHost, adTarget, J24, r):
    """Rows (x,y) of residual r w.r.t. the absolute state [calib(4) | frame_k(6)...] given the relative J and the adjoints."""
    nF = W.nF
    n = 4 + 6 * nF
    h = W.host[W.r_point[r]]
    t = W.r_target[r]
    idx = h + t * nF
    rows = np.zeros((2, n))
    for a in range(2):
        jxi = J24[r, 2 + 6 * a: 8 + 6 * a].astype(np.float64)
        jc = J24[r, 14 + 4 * a: 18 + 4 * a].astype(np.float64)
        rows[a, :4] = jc
        rows[a, 4 + 6 * h: 10 + 6 * h] += adHost[idx] @ jxi
        rows[a, 4 + 6 * t: 10 + 6 * t] += adTarget[idx] @ jxi
    return rows


def dense_normal_equations(W, E, J24, active):
    """Full Gauss-Newton system over [calib | frames | free points] in float64 from the (EF-side) Jacobians."""
    adH, adT = E.adjoints()
    nF = W.nF
    n = 4 + 6 * nF
    free = np.where((W.isFromSensor == 0))[0]
    pidx = -np.ones(W.nP, int)
    pidx[free] = np.arange(len(free))
    N = n + len(free)
    H = np.zeros((N, N))
    b = np.zeros(N)
    for r in np.where(active)[0]:
        rows = np.zeros((2, N))
        rows[:, :n] = abs_jacobian_rows(W, adH, adT, J24, r)
        p = W.r_point[r]
        if pidx[p] >= 0:
            rows[:, n + pidx[p]] = J24[r, 22:24]
        res = J24[r, 0:2].astype(np.float64)
        H += rows.T @ rows
        b += rows.T @ res
    return H, b, n, free, pidx
