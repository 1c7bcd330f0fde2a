// oracle/orc_math.hpp -- TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY PINNED against the reference's own translation units (oracle/_ref/libref.so, oracle/README.md; tests/test_ref_pin*.py).
//
// Small dense maths the reference gets from Eigen3 (un-vendored, README pins >= 3.2.8) and from the
// vendored Sophus 0.9a.  Restated in plain C++ (no Eigen) so that the oracle can be built with g++
// alone.  Every function cites the reference call site / third-party algorithm it follows.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <limits>

namespace orc {

// ---------------------------------------------------------------------------------------------
// 3x3 float inverse, cofactor form -- what Eigen's fixed-size `Matrix3f::inverse()` does
// (Eigen/src/LU/Inverse.h, compute_inverse<MatrixType,ResultType,3>): cofactors of column 0, det as
// their dot product with column 0, every cofactor multiplied by 1/det.
// Reference call sites: CoarseTracker.cpp:100 (Ki[level] = K[level].inverse()),
// HessianBlocks.cpp:188-189 (K.inverse()).  Row-major 3x3.
// ---------------------------------------------------------------------------------------------
inline float cof3(const float* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
inline void inv3f(const float* m, float* out) {
    float c0[3] = {cof3(m, 0, 0), cof3(m, 1, 0), cof3(m, 2, 0)};
    // det = sum_i cofactor(i,0) * m(i,0)
    float det = (c0[0] * m[0] + c0[1] * m[3]) + c0[2] * m[6];
    float invdet = 1.0f / det;
    // result.row(0) = cofactors_col0 * invdet ; result(r,c) = cofactor(c,r) * invdet
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = cof3(m, c, r) * invdet;
}

// ---------------------------------------------------------------------------------------------
// Pivoted LDL^T  (Eigen::LDLT, Eigen/src/Cholesky/LDLT.h, ldlt_inplace<Lower>::unblocked and
// LDLT::_solve_impl): symmetric diagonal pivoting on the biggest |diagonal| of the remaining corner.
// Reference call sites: CoarseTracker.cpp:724 (`Hl.ldlt().solve(-b)`, 8x8 double),
// EnergyFunctional.cpp:743 (`HFinalScaled.ldlt().solve(...)`, (4+6 nF)^2 double).
// Row-major n x n input (only the lower triangle is read, like Eigen).
// ---------------------------------------------------------------------------------------------
inline void ldlt_solve(int n, const double* Ain, const double* bin, double* x) {
    std::vector<double> A(Ain, Ain + (size_t)n * n);
    std::vector<int> tr(n);
    std::vector<double> temp(n);
    auto M = [&](int r, int c) -> double& { return A[(size_t)r * n + c]; };
    for (int k = 0; k < n; ++k) {
        // biggest |diag| in the remaining bottom-right corner
        int idx = k;
        double big = std::fabs(M(k, k));
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(M(i, i)) > big) { big = std::fabs(M(i, i)); idx = i; }
        tr[k] = idx;
        if (idx != k) {
            // symmetric row/col swap restricted to the lower triangle
            for (int c = 0; c < k; ++c) std::swap(M(k, c), M(idx, c));
            for (int r = idx + 1; r < n; ++r) std::swap(M(r, k), M(r, idx));
            std::swap(M(k, k), M(idx, idx));
            for (int i = k + 1; i < idx; ++i) std::swap(M(i, k), M(idx, i));
        }
        const int rs = n - k - 1;
        if (k > 0) {
            for (int c = 0; c < k; ++c) temp[c] = M(c, c) * M(k, c);
            double s = 0;
            for (int c = 0; c < k; ++c) s += M(k, c) * temp[c];
            M(k, k) -= s;
            for (int r = 0; r < rs; ++r) {
                double a = 0;
                for (int c = 0; c < k; ++c) a += M(k + 1 + r, c) * temp[c];
                M(k + 1 + r, k) -= a;
            }
        }
        const double akk = M(k, k);
        if (rs > 0 && std::fabs(akk) > 0.0)
            for (int r = 0; r < rs; ++r) M(k + 1 + r, k) /= akk;
    }
    // solve: x = P^T L^-T D^-1 L^-1 P b
    std::vector<double> y(bin, bin + n);
    for (int k = 0; k < n; ++k) std::swap(y[k], y[tr[k]]);
    for (int r = 0; r < n; ++r) {
        double s = y[r];
        for (int c = 0; c < r; ++c) s -= M(r, c) * y[c];
        y[r] = s;
    }
    const double tol = 1.0 / std::numeric_limits<double>::max();
    for (int i = 0; i < n; ++i) {
        if (std::fabs(M(i, i)) > tol) y[i] /= M(i, i);
        else y[i] = 0;
    }
    for (int r = n - 1; r >= 0; --r) {
        double s = y[r];
        for (int c = r + 1; c < n; ++c) s -= M(c, r) * y[c];
        y[r] = s;
    }
    for (int k = n - 1; k >= 0; --k) std::swap(y[k], y[tr[k]]);
    std::memcpy(x, y.data(), sizeof(double) * n);
}

// ---------------------------------------------------------------------------------------------
// SE(3) with unit-quaternion rotation, Sophus 0.9a layout: data = [qx qy qz qw | tx ty tz]
// (thirdparty/Sophus/sophus/se3.hpp, so3.hpp).  Tangent order = [upsilon(3) | omega(3)].
// ---------------------------------------------------------------------------------------------
struct SE3 {
    double q[4];  // x y z w   (Eigen::Quaternion coeffs order)
    double t[3];
};
const double kSophusEps = 1e-10;  // SophusConstants<double>::epsilon(), sophus.hpp:45-59

inline SE3 se3_identity() { return SE3{{0, 0, 0, 1}, {0, 0, 0}}; }

// Eigen::Quaternion::toRotationMatrix (Eigen/src/Geometry/Quaternion.h); row-major out
inline void quat_to_R(const double* q, double* R) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
    R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
// Eigen quaternion product a*b
inline void quat_mul(const double* a, const double* b, double* o) {
    const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
    const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
    o[3] = aw * bw - ax * bx - ay * by - az * bz;
    o[0] = aw * bx + ax * bw + ay * bz - az * by;
    o[1] = aw * by + ay * bw + az * bx - ax * bz;
    o[2] = aw * bz + az * bw + ax * by - ay * bx;
}
// Eigen QuaternionBase::_transformVector: v + w*uv + q.vec x uv, uv = 2 (q.vec x v)
inline void quat_rot(const double* q, const double* v, double* o) {
    double uv[3] = {q[1] * v[2] - q[2] * v[1], q[2] * v[0] - q[0] * v[2], q[0] * v[1] - q[1] * v[0]};
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    const double c[3] = {q[1] * uv[2] - q[2] * uv[1], q[2] * uv[0] - q[0] * uv[2], q[0] * uv[1] - q[1] * uv[0]};
    o[0] = v[0] + q[3] * uv[0] + c[0];
    o[1] = v[1] + q[3] * uv[1] + c[1];
    o[2] = v[2] + q[3] * uv[2] + c[2];
}
// SO3GroupBase::normalize (so3.hpp:196-202): coeffs /= coeffs.norm().  Eigen sums the four squares of a Matrix<double,4,1> as two
// 2-double packets added lane-wise, then across: (x^2 + z^2) + (y^2 + w^2) (Redux.h packet form, SSE2).
inline void quat_normalize(double* q) {
    const double len = std::sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]));
    for (int i = 0; i < 4; ++i) q[i] /= len;
}
// SE3 operator* = fastMultiply + normalize (se3.hpp:158-166, 268-271; so3.hpp:196-202)
inline SE3 se3_mul(const SE3& a, const SE3& b) {
    SE3 r;
    double rt[3];
    quat_rot(a.q, b.t, rt);
    for (int i = 0; i < 3; ++i) r.t[i] = a.t[i] + rt[i];
    quat_mul(a.q, b.q, r.q);
    quat_normalize(r.q);
    return r;
}
inline SE3 se3_inverse(const SE3& a) {  // se3.hpp:169-173
    SE3 r;
    r.q[0] = -a.q[0]; r.q[1] = -a.q[1]; r.q[2] = -a.q[2]; r.q[3] = a.q[3];
    quat_normalize(r.q);   // so3.hpp:170-172: inverse() = SO3Group(unit_quaternion().conjugate()), the normalising constructor
    double mt[3] = {-a.t[0], -a.t[1], -a.t[2]};
    quat_rot(r.q, mt, r.t);
    return r;
}
// SO3::expAndTheta so3.hpp:342-369 ; SE3::exp se3.hpp:406-427
inline SE3 se3_exp(const double* a) {
    const double* ups = a;
    const double* om = a + 3;
    SE3 r;
    const double theta_sq = om[0] * om[0] + (om[1] * om[1] + om[2] * om[2]);   // omega.squaredNorm(): Eigen's 3-term unrolled reduction = x0 + (x1 + x2)
    const double theta = std::sqrt(theta_sq);
    const double half_theta = 0.5 * theta;
    double imag, real;
    if (theta < kSophusEps) {
        const double theta_po4 = theta_sq * theta_sq;
        imag = 0.5 - (1.0 / 48.0) * theta_sq + (1.0 / 3840.0) * theta_po4;
        real = 1.0 - 0.5 * theta_sq + (1.0 / 384.0) * theta_po4;
    } else {
        const double s = std::sin(half_theta);
        imag = s / theta;
        real = std::cos(half_theta);
    }
    r.q[3] = real; r.q[0] = imag * om[0]; r.q[1] = imag * om[1]; r.q[2] = imag * om[2];
    quat_normalize(r.q);   // `explicit SO3Group(const Quaternion&)` normalises (so3.hpp:596-598)
    // Omega = hat(omega), V
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
            O2[i * 3 + j] = s;
        }
    double V[9];
    if (theta < kSophusEps) {
        quat_to_R(r.q, V);
    } else {
        const double tsq = theta * theta;                      // se3.hpp:418 recomputes theta_sq = theta*theta (not omega.squaredNorm())
        const double c1 = (1.0 - std::cos(theta)) / tsq;
        const double c2 = (theta - std::sin(theta)) / (tsq * theta);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + c1 * O[i] + c2 * O2[i];
    }
    for (int i = 0; i < 3; ++i) r.t[i] = V[i * 3 + 0] * ups[0] + V[i * 3 + 1] * ups[1] + V[i * 3 + 2] * ups[2];
    return r;
}
// SO3::logAndTheta so3.hpp:491-531 ; SE3::log se3.hpp:560-585
inline void se3_log(const SE3& T, double* out) {
    const double sq = T.q[0] * T.q[0] + (T.q[1] * T.q[1] + T.q[2] * T.q[2]);   // vec().squaredNorm(), same reduction shape
    const double n = std::sqrt(sq);
    const double w = T.q[3];
    double two_atan;
    if (n < kSophusEps) {
        two_atan = 2.0 / w - 2.0 * sq / (w * w * w);
    } else if (std::fabs(w) < kSophusEps) {
        two_atan = (w > 0 ? M_PI : -M_PI) / n;
    } else {
        two_atan = 2.0 * std::atan(n / w) / n;
    }
    const double theta = two_atan * n;
    double om[3] = {two_atan * T.q[0], two_atan * T.q[1], two_atan * T.q[2]};
    const double O[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0};
    double O2[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += O[i * 3 + k] * O[k * 3 + j];
            O2[i * 3 + j] = s;
        }
    double c;
    if (std::fabs(theta) < kSophusEps) c = 1.0 / 12.0;
    else c = (1.0 - theta / (2.0 * std::tan(theta / 2.0))) / (theta * theta);
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + c * O2[i * 3 + j]) * T.t[j];
        out[i] = s;
    }
    out[3] = om[0]; out[4] = om[1]; out[5] = om[2];
}
// SE3::Adj se3.hpp:131-139 ; row-major 6x6 : [R, hat(t) R; 0, R]
inline void se3_adj(const SE3& T, double* A) {
    double R[9];
    quat_to_R(T.q, R);
    const double th[9] = {0, -T.t[2], T.t[1], T.t[2], 0, -T.t[0], -T.t[1], T.t[0], 0};
    std::memset(A, 0, sizeof(double) * 36);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            A[i * 6 + j] = R[i * 3 + j];
            A[(i + 3) * 6 + (j + 3)] = R[i * 3 + j];
            double s = 0;
            for (int k = 0; k < 3; ++k) s += th[i * 3 + k] * R[k * 3 + j];
            A[i * 6 + (j + 3)] = s;
        }
}

// AffLight::fromToVecExposure  src/util/NumType.h:149-158
inline void aff_from_to(float expF, float expT, double aF, double bF, double aT, double bT, double* out2) {
    if (expF == 0 || expT == 0) expT = expF = 1;
    const double a = std::exp(aT - aF) * expT / expF;
    const double b = bT - a * bF;
    out2[0] = a; out2[1] = b;
}

}  // namespace orc
