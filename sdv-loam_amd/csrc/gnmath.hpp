// gnmath.hpp -- small dense maths of the Gauss-Newton driver, usable from host code and from HIP kernels.
//
// The reference takes these from Eigen3 / Sophus 0.9a on the CPU:
//   * 8x8 (tracker) and (4+6nF)^2 (back end) `ldlt().solve()`   CoarseTracker.cpp:724, EnergyFunctional.cpp:743
//   * Sophus::SE3d exp / operator* / rotationMatrix             CoarseTracker.cpp:762, :505-506
//   * Matrix3f::inverse()                                       CoarseTracker.cpp:100
//   * AffLight::fromToVecExposure                               src/util/NumType.h:149-158
// Here they are plain functions on C arrays so that the whole LM loop can also run inside one HIP kernel
// (tracker_kernels.hip: k_track) without a host round trip per iteration.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define GN_HD __host__ __device__ __forceinline__

namespace gn {

// pose7 = [qx qy qz qw | tx ty tz]  (Sophus SE3d::data())
struct Pose {
    double q[4];
    double t[3];
};

GN_HD void pose_load(Pose& P, const double* p7) {
    for (int i = 0; i < 4; ++i) P.q[i] = p7[i];
    for (int i = 0; i < 3; ++i) P.t[i] = p7[4 + i];
}
GN_HD void pose_store(const Pose& P, double* p7) {
    for (int i = 0; i < 4; ++i) p7[i] = P.q[i];
    for (int i = 0; i < 3; ++i) p7[4 + i] = P.t[i];
}

// unit quaternion -> row-major rotation matrix (same term grouping as Eigen's toRotationMatrix)
GN_HD void rotation_matrix(const double* q, double* R) {
    const double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
    const double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}

GN_HD void quat_rotate(const double* q, const double* v, double* o) {
    double ax = q[1] * v[2] - q[2] * v[1], ay = q[2] * v[0] - q[0] * v[2], az = q[0] * v[1] - q[1] * v[0];
    ax += ax; ay += ay; az += az;
    o[0] = v[0] + q[3] * ax + (q[1] * az - q[2] * ay);
    o[1] = v[1] + q[3] * ay + (q[2] * ax - q[0] * az);
    o[2] = v[2] + q[3] * az + (q[0] * ay - q[1] * ax);
}

// SO3GroupBase::normalize (so3.hpp:196-202): q /= |q|.  The four squares are added the way Eigen's reduction adds a 4-double vector
// (two 2-lane packets added lane-wise, then across): (x^2 + z^2) + (y^2 + w^2)
GN_HD void quat_normalize(double* q) {
    const double len = sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]));
    for (int i = 0; i < 4; ++i) q[i] /= len;
}

// C = A * B  (group product, then quaternion re-normalisation -- Sophus operator*)
GN_HD Pose compose(const Pose& A, const Pose& B) {
    Pose C;
    double r[3];
    quat_rotate(A.q, B.t, r);
    for (int i = 0; i < 3; ++i) C.t[i] = A.t[i] + r[i];
    const double ax = A.q[0], ay = A.q[1], az = A.q[2], aw = A.q[3];
    const double bx = B.q[0], by = B.q[1], bz = B.q[2], bw = B.q[3];
    C.q[3] = aw * bw - ax * bx - ay * by - az * bz;
    C.q[0] = aw * bx + ax * bw + ay * bz - az * by;
    C.q[1] = aw * by + ay * bw + az * bx - ax * bz;
    C.q[2] = aw * bz + az * bw + ax * by - ay * bx;
    quat_normalize(C.q);
    return C;
}

// sin and cos of one argument.  Host: libm.  Device: Cody-Waite reduction by pi/2 (two-part constant, fused multiply-adds) and the
// fdlibm kernel polynomials on [-pi/4, pi/4] -- within ~1 ulp of libm for |x| < 1e5 (rotation increments are far below that; beyond, the
// reduction loses digits gracefully).  The library's sin / cos carry a Payne-Hanek path for huge arguments whose tables live in the
// private segment: with this, the device-resident LM kernels need no scratch memory.
GN_HD void sincos_d(double x, double* sn, double* cs) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double k = rint(x * 6.36619772367581382433e-01);                      // x * 2/pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    const double z = r * r;
    const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
                      z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double s = r + (r * z) * ps;
    const double c = (1.0 - 0.5 * z) + (z * z) * pc;
    const int q = (int)k & 3;
    const double ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
#else
    *sn = sin(x); *cs = cos(x);
#endif
}

// exp of a twist [upsilon | omega]  (Sophus SE3::exp, se3.hpp:406-427; SO3::expAndTheta so3.hpp:342-369)
GN_HD Pose exp_se3(const double* a) {
    const double wx = a[3], wy = a[4], wz = a[5];
    const double th2 = wx * wx + (wy * wy + wz * wz);       // omega.squaredNorm(): Eigen's unrolled 3-term reduction x0 + (x1 + x2)
    const double th = sqrt(th2);
    const bool tiny = th < 1e-10;
    double im, re;
    if (tiny) {
        const double th4 = th2 * th2;
        im = 0.5 - (1.0 / 48.0) * th2 + (1.0 / 3840.0) * th4;
        re = 1.0 - 0.5 * th2 + (1.0 / 384.0) * th4;
    } else {
        double sh_, ch_;
        sincos_d(0.5 * th, &sh_, &ch_);
        im = sh_ / th;
        re = ch_;
    }
    Pose P;
    P.q[0] = im * wx; P.q[1] = im * wy; P.q[2] = im * wz; P.q[3] = re;
    quat_normalize(P.q);                                    // the SO3Group(Quaternion) constructor normalises (so3.hpp:596-598)
    const double W[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += W[i * 3 + k] * W[k * 3 + j];
            W2[i * 3 + j] = s;
        }
    double V[9];
    if (tiny) {
        rotation_matrix(P.q, V);
    } else {
        double st_, ct_;
        sincos_d(th, &st_, &ct_);
        const double tsq = th * th;                            // se3.hpp:418 recomputes theta*theta
        const double c1 = (1.0 - ct_) / tsq;
        const double c2 = (th - st_) / (tsq * th);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * W[i] + c2 * W2[i];
    }
    for (int i = 0; i < 3; ++i) P.t[i] = V[i * 3] * a[0] + V[i * 3 + 1] * a[1] + V[i * 3 + 2] * a[2];
    return P;
}


// log of a rigid transform -> twist [upsilon | omega]  (Sophus SE3::log se3.hpp:560-585, SO3::logAndTheta so3.hpp:491-531)
GN_HD void log_se3(const Pose& T, double* out) {
    const double sq = T.q[0] * T.q[0] + (T.q[1] * T.q[1] + T.q[2] * T.q[2]);   // vec().squaredNorm(), same reduction shape
    const double nrm = sqrt(sq), w = T.q[3];
    double k;                                    // omega = k * vec(q), theta = k * |vec(q)|
    if (nrm < 1e-10) k = 2.0 / w - 2.0 * sq / (w * w * w);
    else if (fabs(w) < 1e-10) k = (w > 0 ? M_PI : -M_PI) / nrm;
    else k = 2.0 * atan(nrm / w) / nrm;
    const double th = k * nrm;
    const double om[3] = {k * T.q[0], k * T.q[1], k * T.q[2]};
    const double W[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double W2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int q = 0; q < 3; ++q) s += W[i * 3 + q] * W[q * 3 + j];
            W2[i * 3 + j] = s;
        }
    const double c = fabs(th) < 1e-10 ? 1.0 / 12.0 : (1.0 - th / (2.0 * tan(th / 2.0))) / (th * th);
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * W[i * 3 + j] + c * W2[i * 3 + j]) * T.t[j];   // V^-1 t
        out[i] = s;
    }
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}

// inverse of a rigid transform (Sophus SE3::inverse, se3.hpp:169-173)
GN_HD Pose inverse(const Pose& A) {
    Pose R;
    R.q[0] = -A.q[0]; R.q[1] = -A.q[1]; R.q[2] = -A.q[2]; R.q[3] = A.q[3];
    quat_normalize(R.q);                                    // so3.hpp:170-172: the conjugate goes through the normalising constructor
    const double mt[3] = {-A.t[0], -A.t[1], -A.t[2]};
    quat_rotate(R.q, mt, R.t);
    return R;
}

// 6x6 adjoint [R, hat(t) R; 0, R], row-major (Sophus SE3::Adj, se3.hpp:131-139)
GN_HD void adjoint(const Pose& T, double* A) {
    double R[9];
    rotation_matrix(T.q, R);
    const double th[9] = {0, -T.t[2], T.t[1], T.t[2], 0, -T.t[0], -T.t[1], T.t[0], 0};
    for (int i = 0; i < 36; ++i) A[i] = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i * 6 + j] = R[i * 3 + j];
            A[(i + 3) * 6 + (j + 3)] = R[i * 3 + j];
            double s = 0;
            for (int k = 0; k < 3; ++k) s += th[i * 3 + k] * R[k * 3 + j];
            A[i * 6 + (j + 3)] = s;
        }
}

// float 3x3 inverse by cofactors * (1/det), row-major  (what Matrix3f::inverse() evaluates)
GN_HD float cofactor3(const float* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
GN_HD void inverse3f(const float* m, float* out) {
    const float c00 = cofactor3(m, 0, 0), c10 = cofactor3(m, 1, 0), c20 = cofactor3(m, 2, 0);
    const float det = (c00 * m[0] + c10 * m[3]) + c20 * m[6];
    const float invdet = 1.0f / det;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = cofactor3(m, c, r) * invdet;
}

// a = exp(aT-aF) * tT/tF ; b = bT - a*bF      (AffLight::fromToVecExposure)
GN_HD void aff_from_to(float expF, float expT, double aF, double bF, double aT, double bT, double* ab) {
    if (expF == 0 || expT == 0) expT = expF = 1;
    const double a = exp(aT - aF) * expT / expF;
    ab[0] = a;
    ab[1] = bT - a * bF;
}

// In-place LDL^T with symmetric diagonal pivoting (largest |diag| first) + solve.  A is row-major NxN with
// leading dimension lda; only the lower triangle is referenced; A and b are overwritten (b -> x).
// perm_ws / tmp_ws: optional caller-provided work space of MAXN entries each (a kernel passes LDS: no private-segment arrays).
template <int MAXN, bool EXT_WS = false>
GN_HD void ldlt_solve_inplace(int n, double* A, int lda, double* b, int* perm_ws = nullptr, double* tmp_ws = nullptr) {
    int perm_local[EXT_WS ? 1 : MAXN];
    double tmp_local[EXT_WS ? 1 : MAXN];
    int* perm = EXT_WS ? perm_ws : perm_local;
    double* tmp = EXT_WS ? tmp_ws : tmp_local;
#define GN_A(r, c) A[(r) * lda + (c)]
    for (int k = 0; k < n; ++k) {
        int p = k;
        double big = fabs(GN_A(k, k));
        for (int i = k + 1; i < n; ++i) {
            const double v = fabs(GN_A(i, i));
            if (v > big) { big = v; p = i; }
        }
        perm[k] = p;
        if (p != k) {
            for (int c = 0; c < k; ++c) { double s = GN_A(k, c); GN_A(k, c) = GN_A(p, c); GN_A(p, c) = s; }
            for (int r = p + 1; r < n; ++r) { double s = GN_A(r, k); GN_A(r, k) = GN_A(r, p); GN_A(r, p) = s; }
            { double s = GN_A(k, k); GN_A(k, k) = GN_A(p, p); GN_A(p, p) = s; }
            for (int i = k + 1; i < p; ++i) { double s = GN_A(i, k); GN_A(i, k) = GN_A(p, i); GN_A(p, i) = s; }
        }
        if (k > 0) {
            double s = 0;
            for (int c = 0; c < k; ++c) { tmp[c] = GN_A(c, c) * GN_A(k, c); s += GN_A(k, c) * tmp[c]; }
            GN_A(k, k) -= s;
            for (int r = k + 1; r < n; ++r) {
                double a = 0;
                for (int c = 0; c < k; ++c) a += GN_A(r, c) * tmp[c];
                GN_A(r, k) -= a;
            }
        }
        const double d = GN_A(k, k);
        if (k + 1 < n && fabs(d) > 0.0)
            for (int r = k + 1; r < n; ++r) GN_A(r, k) /= d;
    }
    for (int k = 0; k < n; ++k) { double s = b[k]; b[k] = b[perm[k]]; b[perm[k]] = s; }
    for (int r = 0; r < n; ++r) {
        double s = b[r];
        for (int c = 0; c < r; ++c) s -= GN_A(r, c) * b[c];
        b[r] = s;
    }
    for (int i = 0; i < n; ++i) {
        const double d = GN_A(i, i);
        b[i] = (fabs(d) > 5.562684646268003e-309) ? b[i] / d : 0.0;  // 1/DBL_MAX, Eigen's LDLT solve tolerance
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = b[r];
        for (int c = r + 1; c < n; ++c) s -= GN_A(c, r) * b[c];
        b[r] = s;
    }
    for (int k = n - 1; k >= 0; --k) { double s = b[k]; b[k] = b[perm[k]]; b[perm[k]] = s; }
#undef GN_A
}

}  // namespace gn
