"""Known-answer tests that pin the oracle's maths (CPU only).

The reference holds no tests for the hot path (the pin against its own code is tests/test_ref_pin*.py).  The only reference-held
checks that touch it are Sophus' group tests (thirdparty/Sophus/sophus/test_se3.cpp:40-92, tests.hpp:70-110:
exp(log(G)) == G and exp(x) == expm(hat(x))), which the reference does not build; they are re-run here against the
oracle's restatement of SE3::exp/log, on the same group elements and tangent vectors.
"""
import numpy as np
import pytest
from scipy.linalg import expm

SMALL_EPS = 1e-10  # SophusConstants<double>::epsilon()

TANGENTS = [  # test_se3.cpp:67-82
    [0, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 0], [0, 1, 0, 1, 0, 0], [0, -5, 10, 0, 0, 0],
    [-1, 1, 0, 0, 0, 1], [20, -1, 0, -1, 1, 0], [30, 5, -1, 20, -1, 0],
]


def hat(x):
    u, w = x[:3], x[3:]
    M = np.zeros((4, 4))
    M[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    M[:3, 3] = u
    return M


def mat4(orc, p):
    M = np.eye(4)
    M[:3, :3] = orc.se3_matrix(p)
    M[:3, 3] = p[4:]
    return M


def group_elements(orc):  # test_se3.cpp:40-66
    def G(w, t):
        p = orc.se3_exp(np.array([0, 0, 0] + list(w), float))
        p[4:] = t
        return p
    g = [G((0.2, 0.5, 0.0), (0, 0, 0)), G((0.2, 0.5, -1.0), (10, 0, 0)), G((0, 0, 0), (0, 100, 5)),
         G((0, 0, 0.00001), (0, 0, 0)), G((0, 0, 0.00001), (0, -0.00000001, 0.0000000001)),
         G((0, 0, 0.00001), (0.01, 0, 0)), G((np.pi, 0, 0), (4, -5, 0))]
    g.append(orc.se3_mul(orc.se3_mul(G((0.2, 0.5, 0.0), (0, 0, 0)), G((np.pi, 0, 0), (0, 0, 0))), G((-0.2, -0.5, -0.0), (0, 0, 0))))
    g.append(orc.se3_mul(orc.se3_mul(G((0.3, 0.5, 0.1), (2, 0, -7)), G((np.pi, 0, 0), (0, 0, 0))), G((-0.3, -0.5, -0.1), (0, 6, 0))))
    return g


@pytest.mark.parametrize("x", TANGENTS)
def test_sophus_expmap(orc, x):
    """tests.hpp expMapTest: exp(x).matrix() == expm(hat(x)) within 10 eps."""
    x = np.array(x, float)
    assert np.linalg.norm(mat4(orc, orc.se3_exp(x)) - expm(hat(x))) <= 10 * SMALL_EPS * max(1.0, np.linalg.norm(expm(hat(x))))


def test_sophus_explog(orc):
    """tests.hpp expLogTest: G == exp(log(G))."""
    for g in group_elements(orc):
        T1 = mat4(orc, g)
        T2 = mat4(orc, orc.se3_exp(orc.se3_log(g)))
        assert np.linalg.norm(T1 - T2) <= SMALL_EPS * max(1.0, np.linalg.norm(T1))


def test_se3_group_axioms(orc):
    gs = group_elements(orc)
    for a in gs:
        inv = orc.se3_inverse(a)
        assert np.allclose(mat4(orc, orc.se3_mul(a, inv)), np.eye(4), atol=1e-9)
        for b in gs[:4]:
            assert np.allclose(mat4(orc, orc.se3_mul(a, b)), mat4(orc, a) @ mat4(orc, b), atol=1e-9 * np.linalg.norm(mat4(orc, a)))
        # adjoint: T exp(x) T^-1 == exp(Adj x)
        x = np.array([0.1, -0.2, 0.3, 0.02, 0.01, -0.03])
        lhs = mat4(orc, a) @ expm(hat(x)) @ np.linalg.inv(mat4(orc, a))
        rhs = expm(hat(orc.se3_adj(a) @ x))
        assert np.allclose(lhs, rhs, atol=1e-8 * max(1.0, np.abs(lhs).max()))


def test_se3_exp_matches_numpy_mirror(orc):
    from sdv_loam_amd import synthetic as syn
    rng = np.random.default_rng(0)
    for _ in range(50):
        x = rng.normal(0, 0.3, 6)
        assert np.allclose(orc.se3_exp(x), syn.se3_exp_np(x), atol=1e-14)


def test_ldlt_solve(orc):
    rng = np.random.default_rng(1)
    for n in (6, 7, 8, 52):
        A = rng.normal(size=(n, 2 * n))
        A = A @ A.T + 1e-3 * np.eye(n)
        # wide dynamic range like the tracker's scaled H (SCALE_A=10, SCALE_B=1000)
        s = 10.0 ** rng.uniform(-2, 3, n)
        A = A * s[:, None] * s[None, :]
        b = rng.normal(size=n)
        x = orc.ldlt_solve(A, b)
        xr = np.linalg.solve(A, b)
        assert np.allclose(x, xr, rtol=1e-7, atol=1e-12 * np.abs(xr).max())


def test_inv3f_and_makeK(orc):
    from sdv_loam_amd import synthetic as syn
    T = orc.OracleTracker(1241, 376, 4)
    T.makeK(**syn.KITTI00)
    fx, fy, cx, cy = syn.level_intrinsics(syn.KITTI00, 4)
    for l in range(4):
        k4, Ki = T.get_K(l)
        assert np.array_equal(k4, np.array([fx[l], fy[l], cx[l], cy[l]], np.float32))
        K = np.array([[fx[l], 0, cx[l]], [0, fy[l], cy[l]], [0, 0, 1]], np.float64)
        assert np.allclose(Ki, np.linalg.inv(K), rtol=3e-7, atol=1e-9)
    # reference fact (SURVEY.md "three facts" #3): cx_l = (cx0+0.5)/2^l - 0.5, fx_l = fx0/2^l
    assert np.isclose(cx[2], (607.1928 + 0.5) / 4 - 0.5, rtol=1e-6)


def test_interp33_truncation_and_weights(orc):
    rng = np.random.default_rng(2)
    w, h = 17, 9
    img = rng.normal(size=(h, w, 3)).astype(np.float32)
    out = np.zeros(3, np.float32)
    for _ in range(100):
        x = np.float32(rng.uniform(0, w - 1.001))
        y = np.float32(rng.uniform(0, h - 1.001))
        orc.lib().orc_interp33(img.reshape(-1), x, y, w, out)
        ix, iy = int(x), int(y)
        dx, dy = float(x) - ix, float(y) - iy
        ref = ((1 - dx) * (1 - dy) * img[iy, ix] + dx * (1 - dy) * img[iy, ix + 1]
               + (1 - dx) * dy * img[iy + 1, ix] + dx * dy * img[iy + 1, ix + 1])
        assert np.allclose(out, ref, rtol=1e-5, atol=1e-5)
    # integer coordinates return the pixel itself
    orc.lib().orc_interp33(img.reshape(-1), 5.0, 3.0, w, out)
    assert np.array_equal(out, img[3, 5])
