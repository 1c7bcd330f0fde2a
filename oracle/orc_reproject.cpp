// oracle/orc_reproject.cpp -- TEST INFRASTRUCTURE ONLY (CPU oracle). PARITY PINNED against the reference's own translation units (oracle/_ref/libref.so, oracle/README.md; tests/test_ref_pin*.py).
//
// Plain C++ restatement of the per-candidate work of the reference's Reprojector, SURVEY.md section 8f row 2:
//   Reprojector::reprojectPoint            src/FullSystem/Reprojector.cpp:602-616   (projection into the new frame, grid cell)
//   Reprojector::pointQualityComparator    src/FullSystem/Reprojector.cpp:186-194   (the value it compares)
//   Reprojector::findMatchDirect           src/FullSystem/Reprojector.cpp:236-291
//   getWarpMatrixAffine / getBestSearchLevel / warpAffine   :14-79
//   createPatchFromPatchWithBorder         :338-347
//   align1D / align2D                      :349-447 / :449-545
//   pixelFrame2UnitFrame ... pointRef2PixelCur   :547-600
// The control flow around it (grid of candidate lists, per-cell sort, random cell order, first success per cell, stop after
// 0.8*setting_desiredImmatureDensity matches; :117-156,196-234) stays on the host: findMatchDirect is a pure function of
// (point, reference frame, current frame), so evaluating it for EVERY candidate up front and letting the host pick gives the
// reference's overlap_pts exactly.
//
// The reference cannot be compiled here and has no tests for this path (oracle/README.md): pinned by tests/test_oracle_reproject.py.
// Arithmetic follows the reference: double where it uses Eigen::Vector3d / Matrix2d / SE3, float where it uses float, mixed
// expressions promoted like C++ does (e.g. `(1.0-subpix_x)*(1.0-subpix_y)` is a double product rounded to float), sequential float
// accumulation in pixel order, uint8 patches (float -> uint8 truncation).  Build: -ffp-contract=off.
//
// One deliberate deviation: warpAffine returns early when the inverse affine matrix is NaN (:64-68) and findMatchDirect then
// aligns against whatever the previous candidate left in patch_with_border_ -- stale state that depends on the evaluation order.
// Here (and in the HIP kernel) that candidate simply fails.
#include "orc_math.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace orc {

static const int kRpMaxLvl = 6;

struct RpFrame {
    SE3 camToWorld;
    std::vector<float> dI;   // level-0 AoS {I,dx,dy}
    float exposure = 1.f;
    double a = 0, b = 0;     // shell->aff_g2l
};

struct Reproj {
    int levels = 1;
    int w[kRpMaxLvl], h[kRpMaxLvl];
    double K[9], Kinv[9];    // Matrix3d K_ and K_.inverse() (closed-form cofactor inverse, Eigen/src/LU/Inverse.h)
    std::vector<RpFrame> frames;
    SE3 cur_camToWorld;
    std::vector<float> cur_dIp[kRpMaxLvl];
    float cur_exposure = 1.f;
    double cur_a = 0, cur_b = 0;
};

static inline double cof3d(const double* m, int i, int j) {
    const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return m[i1 * 3 + j1] * m[i2 * 3 + j2] - m[i1 * 3 + j2] * m[i2 * 3 + j1];
}
static void inv3d(const double* m, double* out) {
    const double c0[3] = {cof3d(m, 0, 0), cof3d(m, 1, 0), cof3d(m, 2, 0)};
    const double det = (c0[0] * m[0] + c0[1] * m[3]) + c0[2] * m[6];
    const double invdet = 1.0 / det;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) out[r * 3 + c] = cof3d(m, c, r) * invdet;
}
static inline void mat3_vec(const double* M, const double* v, double* o) {
    for (int r = 0; r < 3; ++r) o[r] = (M[3 * r] * v[0] + M[3 * r + 1] * v[1]) + M[3 * r + 2] * v[2];
}
static inline void se3_point(const SE3& T, const double* p, double* o) {   // SE3 * point: so3 * p + t (se3.hpp:257-259)
    double r[3];
    quat_rot(T.q, p, r);
    for (int i = 0; i < 3; ++i) o[i] = r[i] + T.t[i];
}

// pixelFrame2PointWorld (:554-561); `1/point->idepth` is a float division
static void point_world(const Reproj* R, float u, float v, float idepth, const SE3& hostCamToWorld, double* ptWorld) {
    const double pixelRef[3] = {(double)u, (double)v, 1.0};
    double Ki[3];
    mat3_vec(R->Kinv, pixelRef, Ki);
    const double s = (double)(1 / idepth);
    const double ptRef[3] = {Ki[0] * s, Ki[1] * s, Ki[2] * s};
    se3_point(hostCamToWorld, ptRef, ptWorld);
}
// pointWorld2PixelFrame (:563-570)
static void world_to_pixel(const Reproj* R, const SE3& frameCamToWorld, const double* ptWorld, double* pixel) {
    const SE3 w2c = se3_inverse(frameCamToWorld);
    double pc[3];
    se3_point(w2c, ptWorld, pc);
    pc[0] = pc[0] / pc[2]; pc[1] = pc[1] / pc[2]; pc[2] = pc[2] / pc[2];
    mat3_vec(R->K, pc, pixel);
}
// pointRef2PixelCur (:586-593)
static void ref_to_pixel_cur(const Reproj* R, const SE3& T_cur_ref, const double* ptRef, double* px2) {
    double pc[3], pix[3];
    se3_point(T_cur_ref, ptRef, pc);
    pc[0] = pc[0] / pc[2]; pc[1] = pc[1] / pc[2]; pc[2] = pc[2] / pc[2];
    mat3_vec(R->K, pc, pix);
    px2[0] = pix[0]; px2[1] = pix[1];
}
static inline bool in_frame(const Reproj* R, int x, int y, int boundary) {   // isInFrame (:320-326)
    return x >= boundary && x < R->w[0] - boundary && y >= boundary && y < R->h[0] - boundary;
}
static inline float interp_I(const float* mat, float x, float y, int width) {   // getInterpolatedElement33(...)[0]
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float* bp = mat + 3 * (ix + iy * width);
    return ((dxdy * bp[3 + 3 * width] + (dy - dxdy) * bp[3 * width]) + (dx - dxdy) * bp[3]) + (1 - dx - dy + dxdy) * bp[0];
}

// align2D (:449-545).  Returns converged; px (double) is written like `cur_px_estimate << u, v` (not on the NaN return)
static bool align2d(const float* cur_img, int wl, int hl, const uint8_t* pwb, const uint8_t* patch, int n_iter, double* px, const float* affLL) {
    const int half = 4, ps = 8, ref_step = ps + 2;
    bool converged = false;
    float dxs[64], dys[64];
    float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int k = 0;
    for (int y = 0; y < ps; ++y) {
        const uint8_t* it = pwb + (y + 1) * ref_step + 1;
        for (int x = 0; x < ps; ++x, ++it, ++k) {
            float J[3];
            J[0] = (float)(0.5 * (it[1] - it[-1]));
            J[1] = (float)(0.5 * (it[ref_step] - it[-ref_step]));
            J[2] = 1;
            dxs[k] = J[0]; dys[k] = J[1];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) H[3 * r + c] += J[r] * J[c];
        }
    }
    float Hinv[9];
    inv3f(H, Hinv);
    float mean_diff = 0;
    float u = (float)px[0], v = (float)px[1];
    const float min_update_squared = (float)(0.03 * 0.03);
    float update[3] = {0, 0, 0};
    for (int iter = 0; iter < n_iter; ++iter) {
        const int u_r = (int)std::floor(u), v_r = (int)std::floor(v);
        if (u_r < half || v_r < half || u_r >= wl - half || v_r >= hl - half) break;
        if (std::isnan(u) || std::isnan(v)) return false;
        const float sx = u - u_r, sy = v - v_r;
        const float wTL = (float)((1.0 - sx) * (1.0 - sy));
        const float wTR = (float)(sx * (1.0 - sy));
        const float wBL = (float)((1.0 - sx) * sy);
        const float wBR = sx * sy;
        float Jres[3] = {0, 0, 0};
        k = 0;
        for (int y = 0; y < ps; ++y) {
            const float* it = cur_img + 3 * ((size_t)(v_r + y - half) * wl + u_r - half);
            for (int x = 0; x < ps; ++x, it += 3, ++k) {
                const float sp = ((wTL * it[0] + wTR * it[3]) + wBL * it[3 * wl]) + wBR * it[3 * wl + 3];
                const float res = (sp - (float)(affLL[0] * patch[k] + affLL[1])) + mean_diff;
                Jres[0] -= res * dxs[k];
                Jres[1] -= res * dys[k];
                Jres[2] -= res;
            }
        }
        for (int r = 0; r < 3; ++r) update[r] = (Hinv[3 * r] * Jres[0] + Hinv[3 * r + 1] * Jres[1]) + Hinv[3 * r + 2] * Jres[2];
        u += update[0];
        v += update[1];
        mean_diff += update[2];
        if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = true; break; }
    }
    px[0] = u; px[1] = v;
    return converged;
}

// align1D (:349-447)
static bool align1d(const float* cur_img, int wl, int hl, const float* dir, const uint8_t* pwb, const uint8_t* patch, int n_iter, double* px,
                    const float* affLL) {
    const int half = 4, ps = 8, ref_step = ps + 2;
    bool converged = false;
    float dvs[64];
    float H[4] = {0, 0, 0, 0};
    int k = 0;
    for (int y = 0; y < ps; ++y) {
        const uint8_t* it = pwb + (y + 1) * ref_step + 1;
        for (int x = 0; x < ps; ++x, ++it, ++k) {
            float J[2];
            J[0] = (float)(0.5 * (dir[0] * (it[1] - it[-1]) + dir[1] * (it[ref_step] - it[-ref_step])));
            J[1] = 1;
            dvs[k] = J[0];
            for (int r = 0; r < 2; ++r)
                for (int c = 0; c < 2; ++c) H[2 * r + c] += J[r] * J[c];
        }
    }
    // Matrix2f::inverse(): [d -b; -c a] * (1/det)
    const float det = H[0] * H[3] - H[1] * H[2];
    const float invdet = 1.0f / det;
    const float Hinv[4] = {H[3] * invdet, -H[1] * invdet, -H[2] * invdet, H[0] * invdet};
    float mean_diff = 0;
    float u = (float)px[0], v = (float)px[1];
    const float min_update_squared = (float)(0.03 * 0.03);
    float update[2] = {0, 0};
    for (int iter = 0; iter < n_iter; ++iter) {
        const int u_r = (int)std::floor(u), v_r = (int)std::floor(v);
        if (u_r < half || v_r < half || u_r >= wl - half || v_r >= hl - half) break;
        if (std::isnan(u) || std::isnan(v)) return false;
        const float sx = u - u_r, sy = v - v_r;
        const float wTL = (float)((1.0 - sx) * (1.0 - sy));
        const float wTR = (float)(sx * (1.0 - sy));
        const float wBL = (float)((1.0 - sx) * sy);
        const float wBR = sx * sy;
        float Jres[2] = {0, 0};
        k = 0;
        for (int y = 0; y < ps; ++y) {
            const float* it = cur_img + 3 * ((size_t)(v_r + y - half) * wl + u_r - half);
            for (int x = 0; x < ps; ++x, it += 3, ++k) {
                const float sp = ((wTL * it[0] + wTR * it[3]) + wBL * it[3 * wl]) + wBR * it[3 * wl + 3];
                const float res = (sp - (float)(affLL[0] * patch[k] + affLL[1])) + mean_diff;
                Jres[0] -= res * dvs[k];
                Jres[1] -= res;
            }
        }
        update[0] = Hinv[0] * Jres[0] + Hinv[1] * Jres[1];
        update[1] = Hinv[2] * Jres[0] + Hinv[3] * Jres[1];
        u += update[0] * dir[0];
        v += update[0] * dir[1];
        mean_diff += update[1];
        if (update[0] * update[0] + update[1] * update[1] < min_update_squared) { converged = true; break; }
    }
    px[0] = u; px[1] = v;
    return converged;
}

// findMatchDirect (:236-291) for one candidate; ref = the frame the patch is taken from (pt->host when the window has > 2 frames)
static bool find_match(const Reproj* R, float u, float v, float idepth, const RpFrame& host, const RpFrame& ref, int type, double* px_cur,
                       int* level_out) {
    double ab[2];
    aff_from_to(ref.exposure, R->cur_exposure, ref.a, ref.b, R->cur_a, R->cur_b, ab);
    const float affLL[2] = {(float)ab[0], (float)ab[1]};
    double ptWorld[3], ptRef[3], pixelRef[3];
    point_world(R, u, v, idepth, host.camToWorld, ptWorld);
    {   // pointWorld2PointFrame (:572-577)
        const SE3 w2r = se3_inverse(ref.camToWorld);
        se3_point(w2r, ptWorld, ptRef);
    }
    world_to_pixel(R, ref.camToWorld, ptWorld, pixelRef);
    const double px[2] = {pixelRef[0], pixelRef[1]};
    if (!in_frame(R, (int)px[0], (int)px[1], 4 + 2)) return false;

    // getWarpMatrixAffine (:14-36)
    const SE3 T_cur_ref = se3_mul(se3_inverse(R->cur_camToWorld), ref.camToWorld);
    const int hp = 5;
    double A[4];
    {
        const double pdu[3] = {px[0] + hp, px[1] + 0, 1.0}, pdv[3] = {px[0] + 0, px[1] + hp, 1.0};
        double xdu[3], xdv[3];
        mat3_vec(R->Kinv, pdu, xdu);
        mat3_vec(R->Kinv, pdv, xdv);
        const double su = ptRef[2] / xdu[2], sv = ptRef[2] / xdv[2];
        for (int i = 0; i < 3; ++i) { xdu[i] *= su; xdv[i] *= sv; }
        double pc[2], pu[2], pv[2];
        ref_to_pixel_cur(R, T_cur_ref, ptRef, pc);
        ref_to_pixel_cur(R, T_cur_ref, xdu, pu);
        ref_to_pixel_cur(R, T_cur_ref, xdv, pv);
        A[0] = (pu[0] - pc[0]) / hp; A[2] = (pu[1] - pc[1]) / hp;   // col 0
        A[1] = (pv[0] - pc[0]) / hp; A[3] = (pv[1] - pc[1]) / hp;   // col 1
    }
    // getBestSearchLevel (:38-51)
    int lvl = 0;
    {
        double D = A[0] * A[3] - A[1] * A[2];
        while (D > 3.0 && lvl < R->levels - 1) { lvl += 1; D *= 0.25; }
    }
    if (level_out) *level_out = lvl;
    // warpAffine (:53-79) with halfpatch_size_+1 = 5
    uint8_t pwb[100], patch[64];
    {
        const double det = A[0] * A[3] - A[1] * A[2];
        const double invdet = 1.0 / det;
        const float Ai[4] = {(float)(A[3] * invdet), (float)(-A[1] * invdet), (float)(-A[2] * invdet), (float)(A[0] * invdet)};
        if (std::isnan(Ai[0])) return false;   // deviation, see the header comment
        const float prx = (float)px[0], pry = (float)px[1];
        int k = 0;
        for (int y = 0; y < 10; ++y)
            for (int x = 0; x < 10; ++x, ++k) {
                float ppx = (float)(x - 5), ppy = (float)(y - 5);
                ppx *= (1 << lvl); ppy *= (1 << lvl);
                const float qx = (Ai[0] * ppx + Ai[1] * ppy) + prx;
                const float qy = (Ai[2] * ppx + Ai[3] * ppy) + pry;
                if (qx < 0 || qy < 0 || qx >= R->w[0] - 1 || qy >= R->h[0] - 1) pwb[k] = 0;
                else pwb[k] = (uint8_t)(int)interp_I(ref.dI.data(), qx, qy, R->w[0]);
            }
    }
    for (int y = 1; y < 9; ++y)   // createPatchFromPatchWithBorder (:338-347)
        for (int x = 0; x < 8; ++x) patch[(y - 1) * 8 + x] = pwb[y * 10 + 1 + x];

    double pxs[2] = {px_cur[0] / (1 << lvl), px_cur[1] / (1 << lvl)};
    bool success;
    const float* cur_img = R->cur_dIp[lvl].data();
    if (type == 1) {   // EDGELET (:275-284)
        const float* d = ref.dI.data() + 3 * (size_t)(int)(px[0] + px[1] * R->w[0]);
        double g[2] = {(double)d[1], (double)d[2]};
        {   // Eigen 3.3 MatrixBase::normalize(): divide only when the squared norm is > 0
            const double z = g[0] * g[0] + g[1] * g[1];
            if (z > 0) { const double n = std::sqrt(z); g[0] /= n; g[1] /= n; }
        }
        double dc[2] = {A[0] * g[0] + A[1] * g[1], A[2] * g[0] + A[3] * g[1]};
        {
            const double z = dc[0] * dc[0] + dc[1] * dc[1];
            if (z > 0) { const double n = std::sqrt(z); dc[0] /= n; dc[1] /= n; }
        }
        const float dir[2] = {(float)dc[0], (float)dc[1]};
        success = align1d(cur_img, R->w[lvl], R->h[lvl], dir, pwb, patch, 10, pxs, affLL);
    } else {
        success = align2d(cur_img, R->w[lvl], R->h[lvl], pwb, patch, 10, pxs, affLL);
    }
    px_cur[0] = pxs[0] * (1 << lvl);
    px_cur[1] = pxs[1] * (1 << lvl);
    return success;
}

}  // namespace orc

using namespace orc;

extern "C" {
// hook with the signature of oracle/ref_glue.cpp (tests/test_ref_pin.py)
void orc_kat_interp31_reproject(const float* img3, int width, int n, const float* x, const float* y, float* out) {
    for (int i = 0; i < n; ++i) out[i] = interp_I(img3, x[i], y[i], width);
}


void* orc_rp_create(int w0, int h0, int levels) {
    Reproj* R = new Reproj();
    R->levels = levels;
    for (int l = 0; l < levels; ++l) { R->w[l] = w0 >> l; R->h[l] = h0 >> l; }
    return R;
}
void orc_rp_destroy(void* h) { delete (Reproj*)h; }
void orc_rp_set_calib(void* h, float fx, float fy, float cx, float cy) {   // Reprojector::Reprojector (:81-87)
    Reproj* R = (Reproj*)h;
    const double K[9] = {(double)fx, 0, (double)cx, 0, (double)fy, (double)cy, 0, 0, 1};
    std::memcpy(R->K, K, sizeof(K));
    inv3d(R->K, R->Kinv);
}
void orc_rp_set_frame(void* h, int idx, const double* camToWorld7, const float* dI_aos3, float exposure, double a, double b) {
    Reproj* R = (Reproj*)h;
    if ((int)R->frames.size() <= idx) R->frames.resize(idx + 1);
    RpFrame& f = R->frames[idx];
    std::memcpy(f.camToWorld.q, camToWorld7, 32); std::memcpy(f.camToWorld.t, camToWorld7 + 4, 24);
    f.dI.assign(dI_aos3, dI_aos3 + (size_t)3 * R->w[0] * R->h[0]);
    f.exposure = exposure; f.a = a; f.b = b;
}
void orc_rp_set_cur_pose(void* h, const double* camToWorld7, float exposure, double a, double b) {
    Reproj* R = (Reproj*)h;
    std::memcpy(R->cur_camToWorld.q, camToWorld7, 32); std::memcpy(R->cur_camToWorld.t, camToWorld7 + 4, 24);
    R->cur_exposure = exposure; R->cur_a = a; R->cur_b = b;
}
void orc_rp_set_cur_level(void* h, int lvl, const float* dIp_aos3) {
    Reproj* R = (Reproj*)h;
    R->cur_dIp[lvl].assign(dIp_aos3, dIp_aos3 + (size_t)3 * R->w[lvl] * R->h[lvl]);
}
// reprojectPoint (:602-616) for n points + the value pointQualityComparator (:186-194) sorts by
void orc_rp_project(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, double* px2, int* cell,
                    float* quality) {
    const Reproj* R = (const Reproj*)h;
    const int cell_size = 25;
    const int n_cols = (int)std::ceil((double)R->w[0] / cell_size);
    for (int i = 0; i < n; ++i) {
        const RpFrame& host = R->frames[host_idx[i]];
        double ptWorld[3], pix[3];
        point_world(R, u[i], v[i], idepth[i], host.camToWorld, ptWorld);
        world_to_pixel(R, R->cur_camToWorld, ptWorld, pix);
        px2[2 * i] = pix[0]; px2[2 * i + 1] = pix[1];
        if (in_frame(R, (int)pix[0], (int)pix[1], 8)) cell[i] = (int)(pix[1] / cell_size) * n_cols + (int)(pix[0] / cell_size);
        else cell[i] = -1;
        const float* d = host.dI.data() + 3 * (size_t)(int)(v[i] * R->w[0] + u[i]);
        quality[i] = std::sqrt(d[1] * d[1] + d[2] * d[2]);
    }
}
void orc_rp_find_match(void* h, int n, const float* u, const float* v, const float* idepth, const int* host_idx, const int* ref_idx,
                       const int* type, double* px2_io, int* success, int* level) {
    const Reproj* R = (const Reproj*)h;
    for (int i = 0; i < n; ++i) {
        int lvl = -1;
        success[i] = find_match(R, u[i], v[i], idepth[i], R->frames[host_idx[i]], R->frames[ref_idx[i]], type[i], px2_io + 2 * i, &lvl) ? 1 : 0;
        if (level) level[i] = lvl;
    }
}

}  // extern "C"
